#!/bin/bash
# zstd decode: ONE round of four big tiles instead of 5.33 tiles on four streams
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
O=gpurun_out/r05_zstd_tiles_big.txt; : > $O
run() { shape="$1"; shift; for tl in "$@"; do set -- $tl; env="CRYO_ZSTD_LANES=$2"; [ $1 != 0 ] && env="$env,CRYO_ZSTD_TILE=$1"
    echo "== zstd_decode $shape : tile $1 lanes $2" >> $O
    python3 profiles/scripts/ab.py --steps 8 --reps 2 --env $env --args "--workload zstd_decode $shape" zdbg >> $O 2>&1; done; }
run "" "0 4" "16704 4" "22272 3" "22272 4" "32944 2" "13456 5"
run "--dist narrow" "0 4" "16704 4" "22272 3"
run "--blocks 131072" "0 4" "16704 4" "32944 4"
run "--blocks 32768" "0 4" "8352 4" "16704 2"
run "--blocks 16384" "0 4" "4176 4" "8352 2"
run "--block-size 1048576 --blocks 8192" "0 4" "2088 4" "2320 4"
run "--block-size 1048576 --blocks 4096" "0 4" "1392 3" "1392 4" "2088 2"
run "--blocks 16384 --level 5" "0 4" "4176 4"
cat $O
