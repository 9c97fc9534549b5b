#!/bin/bash
# round 4: zstd decode tiles in flight, now that k_zchain4 leaves LDS for the other kernels (debug build: CRYO_ZSTD_LANES)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_z5
{
for l in 2 3 4 5 6 8; do
  python3 profiles/scripts/ab.py --steps 8 --args="--workload zstd_decode" --env CRYO_ZSTD_LANES=$l zdebug | sed "s/^/lanes $l: /"
done
for l in 4 6 8; do
  python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --blocks 8192 --block-size 1048576" --env CRYO_ZSTD_LANES=$l zdebug | sed "s/^/1 MiB lanes $l: /"
done
} 2>&1 | tee gpurun_out/r04_z5/out.txt
