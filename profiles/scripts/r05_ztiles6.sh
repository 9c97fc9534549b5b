#!/bin/bash
# zstd decode: calls of ONE tile (up to 12 288 zstd blocks) cut into two or four (debug build, CRYO_ZSTD_TILE)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
O=gpurun_out/r05_zstd_tiles_small.txt; : > $O
run() { shape="$1"; shift; for t in "$@"; do env="CRYO_ZSTD_LANES=4"; [ $t != 0 ] && env="$env,CRYO_ZSTD_TILE=$t"
    echo "== zstd_decode $shape : tile $t" >> $O
    python3 profiles/scripts/ab.py --steps 10 --reps 2 --env $env --args "--workload zstd_decode $shape" zdbg >> $O 2>&1; done; }
run "--blocks 12288" 0 6144 3072
run "--blocks 8192" 0 4096 2048
run "--blocks 6144" 0 3072 1536
run "--blocks 4096" 0 2048 1024
run "--blocks 2048" 0 1024 512
run "--blocks 8192 --dist narrow" 0 4096 2048
run "--block-size 1048576 --blocks 1024" 0 512 256
run "--block-size 1048576 --blocks 512" 0 256 128
echo "== mixed" >> $O
python3 profiles/scripts/ab.py --steps 10 --reps 2 --args "--workload mixed" zdbg >> $O 2>&1
python3 profiles/scripts/ab.py --steps 10 --reps 2 --env CRYO_ZSTD_TILE=2048 --args "--workload mixed" zdbg >> $O 2>&1
cat $O
