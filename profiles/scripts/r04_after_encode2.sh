#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_after_encode
{
echo "== decode only, a streaming kernel over 8.6 GB of another buffer between the steps (not timed)"
CRYO_BENCH_TOUCH=1 CRYO_BENCH_TRACE=1 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | grep -E "steps" | cut -c1-300
echo "== decode only"
CRYO_BENCH_TRACE=1 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | grep -E "steps" | cut -c1-300
} 2>&1 | tee gpurun_out/r04_after_encode/out2.txt
