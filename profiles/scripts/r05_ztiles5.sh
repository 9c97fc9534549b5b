#!/bin/bash
# zstd decode: is "three large tiles and a small one" better than four equal ones?  (debug build, CRYO_ZSTD_TILE overrides the rule)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
O=gpurun_out/r05_zstd_tiles_unequal.txt; : > $O
run() { shape="$1"; shift; for t in "$@"; do env="CRYO_ZSTD_LANES=4"; [ $t != 0 ] && env="$env,CRYO_ZSTD_TILE=$t"
    echo "== zstd_decode $shape : tile $t" >> $O
    python3 profiles/scripts/ab.py --steps 8 --reps 2 --env $env --args "--workload zstd_decode $shape" zdbg >> $O 2>&1; done; }
run "--block-size 1048576 --blocks 8192" 0 1536 2176 2320 2464 2560 2736 3072
run "" 0 17408 18560 19712 21856
run "--block-size 1048576 --blocks 2048" 0 1024 704
cat $O
