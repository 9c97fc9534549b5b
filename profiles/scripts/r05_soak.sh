#!/bin/bash
# round 5 soak on the final tree: differential stress (device encoders == stock libraries, device decoders of stock streams == input,
# every decode path in rotation) and mutated streams (device verdict and bytes == oracle)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r05_soak.txt; : > $out
for seed in 101 102 103; do timeout 700 python3 tests/stress_gpu.py 500 $seed 2>&1 | tail -2 | tee -a $out; done
for seed in 201 202; do timeout 500 python3 tests/stress_gpu.py fuzz 400 $seed 2>&1 | tail -2 | tee -a $out; done
