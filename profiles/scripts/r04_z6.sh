#!/bin/bash
# round 4: zstd decode tile size (frames per tile; 12 288 = one round of k_zchain4) with four tiles in flight
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_z6
{
for t in 6144 8192 9362 10923 12288 13108 16384 21846; do
  python3 profiles/scripts/ab.py --steps 8 --args="--workload zstd_decode" --env CRYO_ZSTD_TILE=$t zdebug | sed "s/^/tile $t: /"
done
} 2>&1 | tee gpurun_out/r04_z6/out.txt
