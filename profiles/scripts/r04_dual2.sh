#!/bin/bash
# which of the two waves of k_lz4_dec_dual is the longer one: barrier waits of block 0 (variant build with CRYO_DUAL_PROF)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_dual2.txt; : > $out
for args in "--blocks 1024" "--block-size 1048576 --blocks 512"; do
  echo "== $args" | tee -a $out
  CRYO_CODEC_LIB=profiles/variants_dualprof.so timeout 300 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --lz4-waves 2 $args 2>&1 | grep "dual\]" | tail -4 | tee -a $out
done
