#!/bin/bash
# index walkers per block for mid-size batches, with the two-wave decoder behind them
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_dual3.txt; : > $out
for args in "--blocks 1024 --lz4-walkers 8" "--blocks 1024 --lz4-walkers 16" "--blocks 1024 --lz4-walkers 32" "--blocks 1024 --lz4-walkers 64" \
            "--blocks 256 --lz4-walkers 16" "--blocks 256 --lz4-walkers 32" "--blocks 256 --lz4-walkers 64" \
            "--blocks 3072 --lz4-walkers 8" "--blocks 3072 --lz4-walkers 16" "--blocks 3072 --lz4-walkers 32" \
            "--block-size 1048576 --blocks 512 --lz4-walkers 32" "--block-size 1048576 --blocks 512 --lz4-walkers 64"; do
  echo "== $args" | tee -a $out
  timeout 300 python3 profiles/scripts/ab.py --prof --steps 30 "--args=$args" prod 2>&1 | tail -1 | tee -a $out
done
