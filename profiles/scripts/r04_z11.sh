#!/bin/bash
# where one zstd frame's 1.75 ms go (and 64 frames'): per-kernel times
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_z11
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
{
for a in "--blocks 1" "--blocks 64" "--blocks 1024" "--blocks 1 --block-size 1048576"; do
  echo "== $a"
  timeout 300 python3 profiles/scripts/ab.py --prof --steps 20 --args="--workload zstd_decode $a" prod
done
} 2>&1 | tee gpurun_out/r04_z11/out.txt
