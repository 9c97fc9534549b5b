#!/bin/bash
# zstd decode of 1 MiB frames: tiles of one k_zchain round, tiles in flight
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
b() { timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 6 --warmup 2 "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for l in 2 3 4; do export CRYO_ZSTD_LANES=$l; echo "lanes $l: 1 MiB x 8192: $(b --block-size 1048576 --blocks 8192) | 1 MiB x 2048: $(b --block-size 1048576 --blocks 2048) | 256 KiB x 32768: $(b --block-size 262144 --blocks 32768) | 512 KiB x 16384: $(b --block-size 524288 --blocks 16384)"; done 2>&1 | tee gpurun_out/r03_z1m.txt
unset CRYO_ZSTD_LANES
echo "default: 65536 x 128 KiB: $(b)"
