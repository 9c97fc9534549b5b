#!/bin/bash
# zstd encoder tests (levels -5..10) on the GPU
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu -k "encode or unsupported or corners" 2>&1 | tail -25 | tee gpurun_out/r02_zenc.txt
