#!/bin/bash
# a second soak on the round's final tree, other seeds
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r05_soak_b.txt; : > $out
for seed in 111 112; do timeout 450 python3 tests/stress_gpu.py 400 $seed 2>&1 | tail -1 >> $out; done
for seed in 211; do timeout 450 python3 tests/stress_gpu.py fuzz 400 $seed 2>&1 | tail -1 >> $out; done
cat $out
