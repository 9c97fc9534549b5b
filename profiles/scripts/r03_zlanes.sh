#!/bin/bash
# zstd decode: tiles in flight (side streams) and splitting batches of few tiles
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
b() { timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 6 --warmup 2 "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for sp in 1 2 4 8; do export CRYO_ZSTD_SPLIT=$sp; echo "split $sp: 65536: $(b) | 16384: $(b --blocks 16384) | 4096: $(b --blocks 4096) | 1 MiB x 8192: $(b --block-size 1048576 --blocks 8192) | 1 MiB x 512: $(b --block-size 1048576 --blocks 512)"; done 2>&1 | tee gpurun_out/r03_zlanes2.txt
