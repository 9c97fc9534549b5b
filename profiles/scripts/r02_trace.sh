#!/bin/bash
cd "$GRAFT_REPO_ROOT"
CRYO_BENCH_TRACE=1 timeout 600 python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>&1 | grep "bench trace" | head -24
