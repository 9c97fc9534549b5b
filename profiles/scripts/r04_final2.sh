#!/bin/bash
# the last change of the round (k_zexec allocated for six waves per SIMD): suite, smoke, the zstd decode line, a short soak
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_final_check2.txt; : > $out
timeout 1700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee -a $out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $out
timeout 400 python3 tests/stress_gpu.py 120 61 2>&1 | tail -1 | tee -a $out
timeout 400 python3 tests/stress_gpu.py fuzz 80 62 2>&1 | tail -1 | tee -a $out
timeout 900 bash profiles/collect.sh r04 zstd_decode > gpurun_out/r04_collect_zstd_decode.log 2>&1; tail -n 2 gpurun_out/r04_collect_zstd_decode.log | cut -c1-300 | tee -a $out
timeout 300 python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-400 | tee -a $out
