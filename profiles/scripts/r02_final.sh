#!/bin/bash
# final check of the round: full GPU suite, smoke, soak (index path forced on, then default), driver-like bench lines
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r02_final.txt; : > $out
timeout 1700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee -a $out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $out
export CRYO_LZ4_INDEX_MIN=0
timeout 400 python3 tests/stress_gpu.py 200 21 2>&1 | tail -3 | tee -a $out
timeout 400 python3 tests/stress_gpu.py fuzz 150 22 2>&1 | tail -3 | tee -a $out
unset CRYO_LZ4_INDEX_MIN
timeout 400 python3 tests/stress_gpu.py 200 23 2>&1 | tail -3 | tee -a $out
timeout 300 python3 tests/stress_gpu.py fuzz 100 24 2>&1 | tail -3 | tee -a $out
timeout 900 python3 bench.py 2>/dev/null | tail -1 | cut -c1-900 | tee -a $out
timeout 900 python3 bench.py --gpus 2 --steps 40 2>/dev/null | tail -1 | cut -c1-600 | tee -a $out
