#!/bin/bash
# phase timing of k_lz4_dec_seq (debug build: -DCRYO_DEBUG, per-phase s_memtime stamps) on the headline batch
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_stats
export CRYO_CODEC_LIB=$(pwd)/profiles/variants_debug.so CRYO_LZ4_STATS=1
timeout 600 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep "lz4 seq" | tail -12 | tee gpurun_out/r03_stats/stats.txt
