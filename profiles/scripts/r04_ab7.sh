#!/bin/bash
# round 4: batches of 256 ... 4096 blocks: the decoder with an output ring of 16 / 32 / 64 KiB (one wave per workgroup; no far
# matches: every source is in LDS) against the production 4 KiB ring
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_ab7
{
for a in "--blocks 1024" "--blocks 512 --block-size 1048576" "--blocks 4096" "--blocks 256" "--blocks 2048 --block-size 1048576"; do
  echo "== $a"
  python3 profiles/scripts/ab.py --prof --steps 30 --args="$a" prod r16k r32k r64k
done
} 2>&1 | tee gpurun_out/r04_ab7/out.txt
