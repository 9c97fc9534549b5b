#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for wl in mixed zstd zstd_decode; do
echo "== --gpus 2 --workload $wl"; timeout 900 python3 bench.py --gpus 2 --workload $wl --steps 3 --warmup 1 --blocks 8192 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700
done
