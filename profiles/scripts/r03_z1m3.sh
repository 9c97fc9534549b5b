#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export CRYO_CODEC_LIB=$(pwd)/profiles/variants_hwprof.so
CRYO_ZSTD_STATS=1 CRYO_ZSTD_LANES=1 timeout 300 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 1 --warmup 0 --block-size 1048576 --blocks 2048 2>&1 | grep -E "handed back because|k_zhufw wave" | cut -c1-260 | head -4
unset CRYO_CODEC_LIB
bash profiles/scripts/r03_check.sh 100 100
bash profiles/scripts/r03_zshapes.sh | tail -22
