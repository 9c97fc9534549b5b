#!/bin/bash
# zstd decode: tiles of equal size, four per round (prod) against tiles of 12 288 zstd blocks (ztold), one box
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
O=gpurun_out/r05_zstd_equal_tiles.txt; : > $O
python3 -m pytest tests -m gpu -x -q -k "zstd or host or stress or mixed or bench" 2>&1 | tail -2 >> $O
for shape in "" "--dist narrow" "--dist int4" "--blocks 131072" "--blocks 32768" "--blocks 20000" "--blocks 16384" "--blocks 13000" "--blocks 16384 --level 5" "--blocks 16384 --level 3" \
   "--block-size 1048576 --blocks 8192" "--block-size 1048576 --blocks 8192 --dist narrow" "--block-size 1048576 --blocks 4096" "--block-size 1048576 --blocks 2048" "--block-size 262144 --blocks 32768" "--block-size 65536 --blocks 65536"; do
  echo "== zstd_decode $shape" >> $O
  python3 profiles/scripts/ab.py --steps 8 --reps 2 --args "--workload zstd_decode $shape" ztold prod >> $O 2>&1
done
echo "== mixed (configs[4])" >> $O
python3 profiles/scripts/ab.py --steps 8 --reps 1 --args "--workload mixed" ztold prod >> $O 2>&1
cat $O
