#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r05_stream_rot.txt; : > $out
for args in "--dist zeros" "--block-size 1048576 --blocks 8192 --dist zeros" "--block-size 1048576 --blocks 8192 --dist narrow" "--block-size 1048576 --blocks 8192 --dist int4" "--dist narrow" "--block-size 1048576 --blocks 6144 --dist zeros" "--block-size 262144 --blocks 32768 --dist zeros"; do
  echo "== bench.py $args" >> $out
  python3 profiles/scripts/ab.py --prof --steps 30 --args "$args" prod rot rotplain >> $out 2>&1
done
cat $out
