#!/bin/bash
# after the equal-tiles rule of the zstd decode pipeline: the last collection again + the zstd decode workload collections
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
bash profiles/scripts/r05_last.sh > gpurun_out/r05_last_inner.log 2>&1
tail -3 gpurun_out/r05_last_inner.log
T=r05
timeout 900 bash profiles/collect.sh $T zstd_decode > gpurun_out/${T}_collect_zstd_decode.log 2>&1; tail -2 gpurun_out/${T}_collect_zstd_decode.log
timeout 900 bash profiles/collect.sh $T zstd > gpurun_out/${T}_collect_zstd.log 2>&1; tail -2 gpurun_out/${T}_collect_zstd.log
timeout 600 python3 bench.py --workload mixed --steps 20 2>/dev/null | tail -1 > gpurun_out/${T}_mixed_bench.json; cut -c1-300 gpurun_out/${T}_mixed_bench.json
ls gpurun_out | head -50
