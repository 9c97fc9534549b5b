#!/bin/bash
# longer differential soak on the final build: encoders against the stock libraries, decoders (paths rotated per round) against
# the inputs; mutated streams against the oracle's verdicts and bytes
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_soak.txt; : > $out
timeout 700 python3 tests/stress_gpu.py 420 ${1:-77} 2>&1 | tail -3 | tee -a $out
timeout 700 python3 tests/stress_gpu.py fuzz 420 ${2:-78} 2>&1 | tail -3 | tee -a $out
timeout 500 python3 tests/stress_gpu.py 240 ${3:-79} 2>&1 | tail -3 | tee -a $out
