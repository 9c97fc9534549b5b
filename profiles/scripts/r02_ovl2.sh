#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
for args in "--workload lz4_decode" "--workload lz4_decode --blocks 16384" "--workload zstd_decode" "--workload zstd_decode --blocks 4096 --block-size 1048576"; do
echo "$args: $(timeout 600 python3 bench.py $args --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['ms_per_step'])")"
done
export CRYO_LZ4_INDEX_MIN=0
timeout 300 python3 tests/stress_gpu.py 120 31 2>&1 | tail -2
timeout 300 python3 tests/stress_gpu.py fuzz 120 32 2>&1 | tail -2
