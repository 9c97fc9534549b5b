#!/bin/bash
# step 1: index kernel + table batches
export CRYO_LZ4_INDEX_MIN=0
timeout 900 python3 -m pytest tests/test_gpu_lz4.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/s1_pytest_idx.log
unset CRYO_LZ4_INDEX_MIN
timeout 600 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_host.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/s1_pytest.log
bash profiles/quick_stats.sh lz4_decode > gpurun_out/s1_quick.txt 2>&1
CRYO_LZ4_INDEX_MIN=99999999 python3 bench.py --workload lz4_decode --no-cpu-baseline --steps 5 --warmup 1 2>&1 | tail -1 | cut -c1-300 > gpurun_out/s1_old.txt
CRYO_LZ4_STATS=1 python3 bench.py --workload lz4_decode --no-cpu-baseline --steps 1 --warmup 0 --blocks 16384 2>&1 | tail -12 > gpurun_out/s1_stats.txt
cat gpurun_out/s1_pytest_idx.log gpurun_out/s1_pytest.log gpurun_out/s1_quick.txt gpurun_out/s1_old.txt gpurun_out/s1_stats.txt
