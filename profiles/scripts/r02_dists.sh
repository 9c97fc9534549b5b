#!/bin/bash
# LZ4 decode on the other synthetic distributions + 1 MiB blocks + small batches (columns: GB/s, roofline fraction, ratio / ms)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r02_lz4_decode_other_distributions.txt; : > $out
for d in narrow int4 random zeros; do
timeout 600 python3 bench.py --dist $d --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$d', d['value'], d['roofline']['frac'], d['config']['compression_ratio'])" >> $out
done
timeout 600 python3 bench.py --blocks 8192 --block-size 1048576 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('wide 8192 x 1 MiB', d['value'], d['roofline']['frac'], d['config']['compression_ratio'])" >> $out
for nb in 4096 16384 32768; do
timeout 600 python3 bench.py --blocks $nb --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('wide $nb x 128 KiB', d['value'], d['roofline']['frac'], d['ms_per_step'])" >> $out
done
cat $out
