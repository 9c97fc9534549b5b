#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export CRYO_CODEC_LIB=$(pwd)/profiles/variants_hwprof.so
CRYO_ZSTD_STATS=1 CRYO_ZSTD_LANES=1 timeout 300 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 1 --warmup 0 --block-size 1048576 --blocks 512 2>&1 | grep -E "^\[rd\]" | sort | cut -c1-230 | head -34
