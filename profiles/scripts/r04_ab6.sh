#!/bin/bash
# round 4, sixth A/B: index pass with 256-byte rings (7 waves per CU) and two walkers per block, against 512-byte rings / one walker
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_ab6
{
python3 profiles/scripts/ab.py --prof prod ix256_d1 ix256_d2
python3 profiles/scripts/ab.py --prof --args="--lz4-walkers 2" prod ix256_d1 ix256_d2
python3 profiles/scripts/ab.py --prof --args="--lz4-walkers 4" prod ix256_d1 ix256_d2
} 2>&1 | tee gpurun_out/r04_ab6/out.txt
