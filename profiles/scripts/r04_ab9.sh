#!/bin/bash
# round 4: index pass v2 (256-byte rings, 64-byte chunks, 8 waves per CU, two walkers per block on a full batch) against v1
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_ab9
{
timeout 900 python3 -m pytest tests/test_gpu_lz4.py -x -q 2>&1 | tail -2
python3 profiles/scripts/ab.py --prof idxv1 prod idxv2d1
python3 profiles/scripts/ab.py --prof --args="--lz4-walkers 1" prod idxv2d1
python3 profiles/scripts/ab.py --prof --args="--lz4-walkers 4" prod idxv2d1
python3 profiles/scripts/ab.py idxv1 prod idxv1 prod
for a in "--blocks 16384" "--blocks 4096" "--blocks 1024" "--blocks 8192 --block-size 1048576" "--blocks 512 --block-size 1048576" "--dist narrow" "--dist random"; do
  echo "== $a"; python3 profiles/scripts/ab.py --prof --args="$a" idxv1 prod
done
} 2>&1 | tee gpurun_out/r04_ab9/out.txt
