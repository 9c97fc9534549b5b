#!/bin/bash
# thresholds after the batches-across-pieces change: two waves per block beyond 3 328 blocks, more walkers than fill the chip once
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_tune1.txt; : > $out
for args in "--blocks 3328" "--blocks 3328 --lz4-waves 1" "--blocks 4096" "--blocks 4096 --lz4-waves 2" "--blocks 6144" "--blocks 6144 --lz4-waves 2" \
            "--blocks 2048" "--blocks 2048 --lz4-walkers 32" "--blocks 4096 --lz4-walkers 32" "--blocks 8192" "--blocks 8192 --lz4-walkers 16" "--blocks 16384 --lz4-walkers 8" \
            "--block-size 1048576 --blocks 2048" "--block-size 1048576 --blocks 2048 --lz4-walkers 32" "--block-size 1048576 --blocks 2048 --lz4-walkers 64" "--block-size 1048576 --blocks 8192 --lz4-walkers 16"; do
  printf "%-58s " "$args" | tee -a $out
  timeout 300 python3 profiles/scripts/ab.py --prof --steps 20 "--args=$args" prod 2>&1 | tail -1 | sed 's/^prod *//' | tee -a $out
done
