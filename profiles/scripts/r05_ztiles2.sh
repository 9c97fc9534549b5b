#!/bin/bash
# zstd decode, 1 MiB frames: tiles of equal size (production: 1 536 frames x 4 in flight = 5.33 tiles for 8 192 frames)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
O=gpurun_out/r05_zstd_tiles_1m.txt; : > $O
for shape in "--block-size 1048576 --blocks 8192" "--block-size 1048576 --blocks 8192 --dist narrow" "--block-size 1048576 --blocks 2048" "--blocks 131072"; do
  for tl in "0 4" "928 4" "928 5" "928 8" "1392 3" "1392 6" "1856 5" "2320 4" "2784 3" "2784 4" "4176 2" "4176 4"; do
    set -- $tl
    case "$shape" in *131072*) case "$1" in 0) ;; 928) set -- 10672 4;; 1392) set -- 10672 6;; 2320) set -- 8352 4;; 2784) set -- 8352 8;; *) continue;; esac;; esac
    env="CRYO_ZSTD_LANES=$2"; [ $1 != 0 ] && env="$env,CRYO_ZSTD_TILE=$1"
    echo "== zstd_decode $shape : tile $1 (0 = production) lanes $2" >> $O
    python3 profiles/scripts/ab.py --steps 8 --reps 2 --env $env --args "--workload zstd_decode $shape" zdbg >> $O 2>&1
  done
done
cat $O
