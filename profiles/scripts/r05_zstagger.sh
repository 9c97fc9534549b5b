#!/bin/bash
# zstd decode: tiles of two sizes alternating per stream and round (debug build) against four equal tiles
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
O=gpurun_out/r05_zstd_stagger.txt; : > $O
run() { shape="$1"; shift; for ts in "$@"; do set -- $ts; env="CRYO_ZSTD_LANES=4"; [ $1 != 0 ] && env="$env,CRYO_ZSTD_TILE=$1,CRYO_ZSTD_STAGGER=$2"
    echo "== zstd_decode $shape : tile $1 stagger $2 %" >> $O
    python3 profiles/scripts/ab.py --steps 8 --reps 2 --env $env --args "--workload zstd_decode $shape" zdbg >> $O 2>&1; done; }
run "" "0 0" "10240 60" "9840 67" "10928 50" "12288 34" "11472 43"
run "--dist narrow" "0 0" "10240 60" "10928 50"
run "--block-size 1048576 --blocks 8192" "0 0" "1280 60" "1376 49" "1536 34"
run "--blocks 16384 --level 5" "0 0" "2560 60"
cat $O
