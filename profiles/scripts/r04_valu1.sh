#!/bin/bash
# fewer vector instructions per batch (votes folded into their compares, lane_runs loop without register copies):
# parity of everything that shares lz4_copy.h, then prod against the build before (variants_nodual.so = HEAD~ kernels)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_valu1.txt; : > $out
timeout 1200 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_zstd.py tests/test_gpu_bench_workloads.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out
for v in ${VARIANTS:-prod nodual prod nodual}; do
  timeout 400 python3 profiles/scripts/ab.py --prof --steps 20 $v 2>&1 | tail -1 | tee -a $out
done
for v in prod nodual; do
  timeout 400 python3 profiles/scripts/ab.py --prof --steps 10 --args="--workload zstd_decode" $v 2>&1 | tail -1 | tee -a $out
done
