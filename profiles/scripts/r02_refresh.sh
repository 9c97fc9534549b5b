#!/bin/bash
# the smaller evidence files of the round, on the final HEAD
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r02_bench_driver_args.json
timeout 400 python3 bench.py --workload mixed 2>/dev/null | tail -1 > gpurun_out/r02_mixed_bench.json
timeout 300 python3 profiles/host_api_rate.py > gpurun_out/r02_host_api.txt 2>&1
CRYO_PIPE_MIN_MB=999999 timeout 300 python3 profiles/host_api_rate.py 2>&1 | sed 's/^/[one-shot path] /' >> gpurun_out/r02_host_api.txt
cut -c1-300 gpurun_out/r02_bench_driver_args.json; cut -c1-300 gpurun_out/r02_mixed_bench.json; cat gpurun_out/r02_host_api.txt
