#!/bin/bash
# round 5: the last, partial round of an LZ4 decode on a low-priority side stream with two waves per block (auto) against
# one wave per block throughout (--lz4-waves 1 = round 4's behaviour for these sizes)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_bench_workloads.py -x -q -m gpu -k "not full_size" 2>&1 | tail -3
out=gpurun_out/r05_lz4_tail_round.txt; : > $out
for args in "--block-size 1048576 --blocks 8192" "--block-size 1048576 --blocks 7168" "--block-size 1048576 --blocks 12800" "--blocks 8192" "--blocks 7000" "--blocks 9216" "--blocks 14336" "--blocks 20480" "--blocks 16384" "--block-size 1048576 --blocks 8192 --dist narrow"; do
  for w in "" "--lz4-waves 1"; do
    echo "== bench.py $args $w" >> $out
    python3 profiles/scripts/ab.py --prof --steps 30 --args "$args $w" prod >> $out 2>&1
  done
done
cat $out
