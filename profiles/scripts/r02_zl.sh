#!/bin/bash
for l in 1 2; do CRYO_ZSTD_LANES=$l python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('lanes $l', j['value'], 'GB/s', j['ms_per_step'], 'ms')"; done
