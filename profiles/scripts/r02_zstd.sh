#!/bin/bash
timeout 1200 python3 -m pytest tests/test_gpu_zstd.py tests/test_gpu_stress.py tests/test_gpu_host.py -x -q -m gpu 2>&1 | tail -4
bash profiles/quick_stats.sh zstd_decode 2>&1 | tail -9
timeout 300 python3 tests/stress_gpu.py 60 21 2>&1 | tail -2
timeout 300 python3 tests/stress_gpu.py fuzz 60 22 2>&1 | tail -2
