#!/bin/bash
# LZ4 few-blocks path: parity, then the small batch shapes with and without it
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_lz4.py -x -q -m gpu 2>&1 | tail -5
out=gpurun_out/r03_lz4_few_blocks.txt; : > $out
b() { timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('%-62s %9.1f GB/s  %8.3f ms' % ('$*', d['value'], d['ms_per_step']))" >> $out; }
for sh in "--block-size 1048576 --blocks 1" "--block-size 1048576 --blocks 16" "--block-size 1048576 --blocks 64" "--blocks 1" "--blocks 16" "--blocks 64" "--blocks 256"; do
  b $sh
  b $sh --lz4-path 2
done
cat $out
