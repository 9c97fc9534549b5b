#!/bin/bash
# the index pass moves between 2.03 and 2.22 ms from process to process (same binary, same box): how wide is the spread, and do
# non-temporal row stores change it?  six processes each, alternating
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r05_index_spread.txt; : > $out
for i in 1 2 3 4 5 6; do
  python3 profiles/scripts/ab.py --prof --steps 30 prod idxnt >> $out 2>&1
done
cat $out
