#!/bin/bash
# zstd decode over batch shapes, levels and distributions (bench.py, outputs verified against the inputs)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r04_zstd_decode_batch_shapes.txt; : > $out
b() { timeout 900 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps ${STEPS:-8} --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); c = d['config']
    print('%-50s %9.1f GB/s  %9.3f ms  ratio %s' % ('$*' or '(65536 x 128 KiB wide, level 1)', d['value'], d['ms_per_step'], c['compression_ratio']))
except Exception as e:
    print('%-50s FAILED %s' % ('$*', e))" >> $out; }
b
b --blocks 16384
b --blocks 4096
b --blocks 1024
b --blocks 64
b --blocks 16
b --blocks 1
b --block-size 1048576 --blocks 8192
b --block-size 1048576 --blocks 512
b --block-size 1048576 --blocks 16
b --block-size 1048576 --blocks 1
b --blocks 16384 --level 3
b --blocks 16384 --level 5
b --blocks 16384 --level 10
b --blocks 16384 --level -5
b --dist narrow
b --dist int4
b --dist random
b --dist zeros
cat $out
timeout 400 python3 bench.py --workload mixed 2>/dev/null | tail -1 | cut -c1-300
