#!/bin/bash
# the last collection of round 5 (after the zstd few-frames path; run again after the literal-section threshold): suite, smoke, zstd batch shapes, crossover, two ranks
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
T=r05
out=gpurun_out/${T}_final_check.txt; : > $out
timeout 1700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee -a $out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $out
timeout 900 python3 bench.py --gpus 2 --steps 20 2>/dev/null | tail -1 | cut -c1-900 | tee -a $out
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-700 | tee -a $out
out=gpurun_out/${T}_zstd_decode_batch_shapes.txt; : > $out
z() { timeout 900 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 8 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); c = d['config']
    print('%-52s %9.1f GB/s  %9.3f ms  frac %.4f  ratio %s' % ('$*' or '(65536 x 128 KiB wide, level 1)', d['value'], d['ms_per_step'], d['roofline']['frac'], c['compression_ratio']))
except Exception as e:
    print('%-52s FAILED %s' % ('$*', e))" >> $out; }
z
z --blocks 16384
z --blocks 4096
z --blocks 1024
z --blocks 64
z --blocks 16
z --blocks 1
z --block-size 1048576 --blocks 8192
z --block-size 1048576 --blocks 512
z --block-size 1048576 --blocks 64
z --block-size 1048576 --blocks 16
z --block-size 1048576 --blocks 1
z --blocks 16384 --level 3
z --blocks 16384 --level 5
z --blocks 16384 --level -5
z --dist narrow
z --dist int4
z --dist random
z --dist zeros
z --block-size 1048576 --blocks 8192 --dist narrow
z --block-size 1048576 --blocks 16 --dist narrow
z --block-size 1048576 --blocks 1 --dist narrow
cat $out
timeout 900 python3 profiles/crossover.py 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/${T}_crossover.txt; tail -60 gpurun_out/${T}_crossover.txt
