#!/bin/bash
timeout 900 python3 -m pytest tests/test_gpu_host.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -4
python3 profiles/host_api_rate.py 2>&1 | tee gpurun_out/r02_host_api.txt
CRYO_PIPE_MIN_MB=999999 python3 profiles/host_api_rate.py 2>&1 | sed 's/^/[one-shot path] /' | tee -a gpurun_out/r02_host_api.txt
