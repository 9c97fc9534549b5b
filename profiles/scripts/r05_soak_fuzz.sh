#!/bin/bash
# second half of the round-5 soak on the final tree (the first half used up one call's limit): mutated streams, device verdict and bytes == oracle
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r05_soak_fuzz.txt; : > $out
for seed in 201 202; do timeout 450 python3 tests/stress_gpu.py fuzz 400 $seed 2>&1 | tail -2 >> $out; done
python3 -m pytest tests/test_gpu_zstd.py -m gpu -x -q -k "threshold" 2>&1 | tail -3 >> $out
cat $out
