#!/bin/bash
# round 5 A/B 1: non-temporal output stores (product) against plain ones (nt0), non-temporal input loads on top (ntl)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r05_nt_ab.txt; : > $out
for args in "" "--dist zeros" "--dist narrow" "--dist int4" "--dist random" "--blocks 1024" "--block-size 1048576 --blocks 8192" "--workload zstd_decode"; do
  echo "== bench.py $args" >> $out
  python3 profiles/scripts/ab.py --prof --steps 30 --args "$args" nt0 prod ntl >> $out 2>&1
done
cat $out
