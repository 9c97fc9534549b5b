#!/bin/bash
# k_zexec at 6 waves per SIMD is prod now: zstd parity, then k_zmat at 7 / 8 waves against it
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_zocc2.txt; : > $out
timeout 1200 python3 -m pytest tests/test_gpu_zstd.py tests/test_gpu_bench_workloads.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out
for v in prod zmat7 zmat8 prod zmat7; do
  timeout 400 python3 profiles/scripts/ab.py --prof --steps 10 --args="--workload zstd_decode" $v 2>&1 | tail -1 | tee -a $out
done
for v in prod zmat7; do
  timeout 400 python3 profiles/scripts/ab.py --steps 10 --args="--workload zstd_decode --block-size 1048576 --blocks 8192" $v 2>&1 | tail -1 | tee -a $out
done
