#!/bin/bash
# zstd decode: tile size and tiles in flight once more, on the round-5 kernels (debug build; production: 12 288 frames x 4)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
O=gpurun_out/r05_zstd_tiles.txt; : > $O
for shape in "" "--dist narrow" "--block-size 1048576 --blocks 8192"; do
  for tl in "0 4" "0 3" "0 5" "0 6" "0 8" "6032 4" "6032 6" "6032 8" "8352 4" "8352 6" "4176 8"; do
    set -- $tl
    env="CRYO_ZSTD_LANES=$2"; [ $1 != 0 ] && env="$env,CRYO_ZSTD_TILE=$1"
    echo "== zstd_decode $shape : tile $1 (0 = production) lanes $2" >> $O
    python3 profiles/scripts/ab.py --steps 10 --reps 2 --env $env --args "--workload zstd_decode $shape" zdbg >> $O 2>&1
  done
done
cat $O
