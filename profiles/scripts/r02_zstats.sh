#!/bin/bash
cd "$GRAFT_REPO_ROOT"
CRYO_ZSTD_STATS=1 timeout 600 python3 bench.py --workload zstd_decode --steps 1 --warmup 0 --blocks 16384 --no-cpu-baseline 2>&1 | grep "zstd pipe" | head -3
