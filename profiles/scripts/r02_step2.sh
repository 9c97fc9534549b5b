#!/bin/bash
# new decoder (index + run-space copy engine): parity tests with the index forced on, then timing
export CRYO_LZ4_INDEX_MIN=0
timeout 900 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/s2_pytest_idx.log
unset CRYO_LZ4_INDEX_MIN
bash profiles/quick_stats.sh lz4_decode > gpurun_out/s2_quick.txt 2>&1
CRYO_LZ4_STATS=1 python3 bench.py --workload lz4_decode --no-cpu-baseline --steps 1 --warmup 0 --blocks 32768 2>&1 | grep "lz4 seq" > gpurun_out/s2_stats.txt
for d in narrow int4 random zeros; do python3 bench.py --workload lz4_decode --no-cpu-baseline --steps 5 --warmup 1 --dist $d 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$d', j['value'], j['roofline']['frac'])"; done > gpurun_out/s2_dists.txt 2>&1
cat gpurun_out/s2_pytest_idx.log gpurun_out/s2_quick.txt gpurun_out/s2_stats.txt gpurun_out/s2_dists.txt
