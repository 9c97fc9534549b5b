#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -3
out=gpurun_out/r05_index_run255.txt; : > $out
for args in "--block-size 1048576 --blocks 8192 --dist zeros" "--block-size 1048576 --blocks 8192 --dist narrow" "--block-size 1048576 --blocks 8192 --dist int4" "--dist narrow" "--dist int4" "--dist zeros" "" "--block-size 1048576 --blocks 8192" "--block-size 1048576 --blocks 512 --dist narrow" "--block-size 1048576 --blocks 64 --dist narrow"; do
  echo "== bench.py $args" >> $out
  python3 profiles/scripts/ab.py --prof --steps 30 --args "$args" prod >> $out 2>&1
done
cat $out
timeout 300 python3 tests/stress_gpu.py 120 71 2>&1 | tail -2
