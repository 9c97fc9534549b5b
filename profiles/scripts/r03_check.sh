#!/bin/bash
# GPU suite, smoke, differential soak and fuzz
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r03_check.txt; : > $out
timeout 1700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee -a $out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $out
timeout 400 python3 tests/stress_gpu.py ${1:-150} 41 2>&1 | tail -3 | tee -a $out
timeout 400 python3 tests/stress_gpu.py fuzz ${2:-150} 42 2>&1 | tail -3 | tee -a $out
