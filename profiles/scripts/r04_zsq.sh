#!/bin/bash
# instruction mix and pipe occupancy of the zstd decode pipeline's kernels (SQ counters, three passes)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
timeout 1200 bash profiles/scripts/pmc_sq.sh r04_zstd_dec zstd_decode > gpurun_out/r04_zsq.log 2>&1
tail -12 gpurun_out/r04_zsq.log
