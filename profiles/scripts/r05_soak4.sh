#!/bin/bash
# a last short soak on the round's last tree (after the zstd tile rule)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r05_soak_c.txt; : > $out
timeout 260 python3 tests/stress_gpu.py 200 121 2>&1 | tail -1 >> $out
timeout 260 python3 tests/stress_gpu.py fuzz 200 221 2>&1 | tail -1 >> $out
cat $out
