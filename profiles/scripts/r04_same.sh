#!/bin/bash
# round 4 diagnostic: how much of the LZ4 decode call is waiting for the compressed input?  Every block decodes block 0's stream
# (input lines and index rows served from cache; sequences, copies and output stores unchanged)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_same
{
python3 profiles/scripts/ab.py --prof --args="--no-verify" prod
python3 profiles/scripts/ab.py --prof --args="--no-verify" --env CRYO_BENCH_SAME_BLOCK=1 prod
python3 profiles/scripts/ab.py --prof --args="--no-verify --blocks 1024" prod
python3 profiles/scripts/ab.py --prof --args="--no-verify --blocks 1024" --env CRYO_BENCH_SAME_BLOCK=1 prod
} 2>&1 | tee gpurun_out/r04_same/out.txt
