#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r05_few_tests.txt
out=gpurun_out/r05_lz4_few_blocks.txt; : > $out
for args in "--block-size 1048576 --blocks 1" "--block-size 1048576 --blocks 4" "--block-size 1048576 --blocks 16" "--block-size 1048576 --blocks 64" "--blocks 1" "--blocks 16" "--blocks 64" "--block-size 1048576 --blocks 1 --dist narrow" "--block-size 1048576 --blocks 16 --dist int4" "--block-size 262144 --blocks 16"; do
  echo "== bench.py $args" >> $out
  python3 profiles/scripts/ab.py --prof --steps 50 --args "$args" latold prod >> $out 2>&1
done
cat $out
timeout 500 python3 tests/stress_gpu.py 200 51 2>&1 | tail -3 | tee gpurun_out/r05_soak_mid.txt
timeout 300 python3 tests/stress_gpu.py fuzz 120 52 2>&1 | tail -3 | tee -a gpurun_out/r05_soak_mid.txt
