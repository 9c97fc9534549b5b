#!/bin/bash
# the one-piece instantiation for one walker per block: parity, headline and automatic settings against the build before
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_piece2.txt; : > $out
timeout 1200 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_bench_workloads.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out
for args in "" "" "--blocks 131072" "--blocks 32768" "--blocks 16384" "--blocks 4096" "--blocks 1024" "--blocks 256" "--block-size 1048576 --blocks 8192" "--block-size 1048576 --blocks 512" "--blocks 16384 --dist narrow"; do
  for v in prod prev; do
    printf "%-44s %-5s " "$args" $v | tee -a $out
    timeout 300 python3 profiles/scripts/ab.py --prof --steps 20 "--args=$args" $v 2>&1 | tail -1 | sed 's/^[a-z]* *//' | tee -a $out
  done
done
