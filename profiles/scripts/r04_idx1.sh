#!/bin/bash
# where the index pass's time goes with several walkers per block: the walk against the hand-over (variant build, CRYO_IDX_PROF)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_idx1.txt; : > $out
for args in "--blocks 1024" "--blocks 64 --lz4-path 2" "--block-size 1048576 --blocks 512" "--blocks 16384"; do
  echo "== $args" | tee -a $out
  CRYO_CODEC_LIB=profiles/variants_${1:-idxprof}.so timeout 300 python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 $args 2>&1 | grep "index\]" | sort -t: -k2 | tail -12 | tee -a $out
done
