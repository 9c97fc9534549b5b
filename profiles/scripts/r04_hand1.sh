#!/bin/bash
# the index pass's hand-over through the idle rings: parity, then prod against the build before (variants_prev.so)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_hand1.txt; : > $out
timeout 1200 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_bench_workloads.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out
for args in "--blocks 1024" "--blocks 256" "--blocks 4096" "--blocks 16384" "--blocks 32768" "--block-size 1048576 --blocks 512" "--block-size 1048576 --blocks 8192" "--block-size 1048576 --blocks 16" "--blocks 16 " "--blocks 4096 --dist narrow" "--blocks 4096 --dist random" "--blocks 4096 --dist int4"; do
  for v in prod prev; do
    printf "%-44s %-5s " "$args" $v | tee -a $out
    timeout 300 python3 profiles/scripts/ab.py --prof --steps 20 "--args=$args" $v 2>&1 | tail -1 | sed 's/^[a-z]* *//' | tee -a $out
  done
done
bash profiles/scripts/r04_idx1.sh idxprof > /dev/null 2>&1; cat gpurun_out/r04_idx1.txt | tee -a $out
