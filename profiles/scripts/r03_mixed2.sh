#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_mixed2
bm() { timeout 600 python3 bench.py --workload mixed --steps 20 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', '| GB/s', d['value'], 'ms', d['ms_per_step'])"; }
bl() { timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', '| GB/s', d['value'], 'ms', d['ms_per_step'], 'ratio', d['config']['compression_ratio'])"; }
{
timeout 900 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -3
bm "mixed default"
CRYO_LZ4_DECODE_PATH=1 bm "mixed, lz4 half on the in-wave parser"
bl --blocks 8192 --accel 50
bl --blocks 8192 --accel 50 --lz4-path 1
bl --blocks 8192 --accel 50 --lz4-walkers 1
bl --blocks 8192 --accel 50 --lz4-walkers 2
bl --blocks 65536 --accel 50
bl --blocks 65536 --accel 50 --lz4-path 1
bl --blocks 16384
bl --blocks 4096
bl --block-size 1048576 --blocks 8192
bl --blocks 16384 --dist random
bl --blocks 16384 --dist random --lz4-path 1
bl --blocks 16384 --dist narrow
bl --blocks 16384 --dist narrow --lz4-path 1
} 2>&1 | tee gpurun_out/r03_mixed2/out.txt
