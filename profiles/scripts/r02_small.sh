#!/bin/bash
timeout 1200 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_stress.py tests/test_gpu_host.py -x -q -m gpu 2>&1 | tail -4
for nb in 4096 16384; do python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --blocks $nb 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('blocks', $nb, 'GB/s', j['value'], 'ms', j['ms_per_step'])"; done
CRYO_LZ4_INDEX_MIN=99999999 python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('in-wave parse 65536: GB/s', j['value'], 'ms', j['ms_per_step'])"
python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('indexed 65536: GB/s', j['value'], 'ms', j['ms_per_step'])"
for d in narrow int4 random zeros; do CRYO_LZ4_INDEX_MIN=99999999 python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 --dist $d 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('in-wave $d', j['value'], j['roofline']['frac'])"; done
