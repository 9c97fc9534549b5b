#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for d in narrow int4; do echo "== zstd_decode --dist $d"; bash profiles/quick_stats.sh zstd_decode --dist $d 2>&1 | tail -12; done > gpurun_out/r05_zstd_narrow_kernels.txt 2>&1
echo "== zstd_decode --dist narrow --block-size 1048576 --blocks 8192" >> gpurun_out/r05_zstd_narrow_kernels.txt
bash profiles/quick_stats.sh zstd_decode --dist narrow --block-size 1048576 --blocks 8192 2>&1 | tail -12 >> gpurun_out/r05_zstd_narrow_kernels.txt
cat gpurun_out/r05_zstd_narrow_kernels.txt
