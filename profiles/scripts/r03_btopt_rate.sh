#!/bin/bash
# zstd encode rate at the optimal-parser levels, and the configs[4] line with the level-22 half encoded on the GPU
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r03_zstd_btopt_rate.txt; : > $out
for spec in "13 131072 2816" "16 131072 2816" "19 131072 2816" "22 131072 2816" "22 16384 8192" "16 1048576 512" "22 1048576 512"; do
  set -- $spec
  timeout 900 python3 bench.py --workload zstd --level $1 --blocks $3 --block-size $2 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('level $1, $3 x $2 bytes: encode %s GB/s, decode %s GB/s, ratio %s' % (c.get('encode_GBps'), c.get('decode_GBps'), c.get('compression_ratio')))" >> $out
done
timeout 900 python3 bench.py --workload mixed --steps 20 --warmup 2 --no-cpu-baseline 2>gpurun_out/r03_mixed.err | tail -1 > gpurun_out/r03_mixed_bench.json
cat $out; cat gpurun_out/r03_mixed_bench.json; tail -3 gpurun_out/r03_mixed.err
