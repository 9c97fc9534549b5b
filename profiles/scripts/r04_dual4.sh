#!/bin/bash
# k_lz4_dec_dual with match space prepared by wave A: parity, rates, barrier waits
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_dual4.txt; : > $out
timeout 900 python3 -m pytest tests/test_gpu_lz4.py -x -q -m gpu 2>&1 | tail -4 | tee -a $out
for args in "--blocks 1024" "--blocks 3072" "--blocks 256" "--block-size 1048576 --blocks 512"; do
  echo "== $args" | tee -a $out
  timeout 300 python3 profiles/scripts/ab.py --prof --steps 30 "--args=$args" prod 2>&1 | tail -1 | tee -a $out
  CRYO_CODEC_LIB=profiles/variants_dualprof.so timeout 300 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 $args 2>&1 | grep "dual\]" | tail -2 | tee -a $out
done
