#!/bin/bash
# the optimal-parser levels on the GPU: bit-exactness against oracle + stock library, time per call
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout ${2:-900} python3 profiles/scripts/r03_btopt_check.py $1 > gpurun_out/r03_zstd_btopt.txt 2>&1
tail -60 gpurun_out/r03_zstd_btopt.txt
