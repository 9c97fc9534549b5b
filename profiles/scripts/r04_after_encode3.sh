#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_after_encode
{
echo "== compress, decompress, decompress again"
CRYO_BENCH_DEC_AGAIN=1 CRYO_BENCH_TRACE=1 python3 bench.py --workload lz4 --no-cpu-baseline --steps 8 --warmup 2 2>&1 | grep -E "bench trace" | cut -c1-300
echo "== the same with zstd as the encoder (level 1), decoded by the zstd pipeline"
CRYO_BENCH_DEC_AGAIN=1 CRYO_BENCH_TRACE=1 python3 bench.py --workload zstd --no-cpu-baseline --steps 4 --warmup 1 2>&1 | grep -E "bench trace" | cut -c1-300
} 2>&1 | tee gpurun_out/r04_after_encode/out3.txt
