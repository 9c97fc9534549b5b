#!/bin/bash
# zstd encode rate by level (the hash-chain levels after the miss-run skip)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r03_zstd_levels.txt; : > $out
for lvl in 1 3 5 6 7 8 9 10; do
  timeout 300 python3 bench.py --workload zstd --level $lvl --blocks 4096 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('level $lvl 128KiB x4096: encode %s GB/s decode %s GB/s ratio %s' % (c.get('encode_GBps'), c.get('decode_GBps'), c.get('compression_ratio')))" >> $out
done
for lvl in 1 6 10 12; do
  timeout 300 python3 bench.py --workload zstd --level $lvl --blocks 512 --block-size 1048576 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('level $lvl 1MiB x512: encode %s GB/s decode %s GB/s ratio %s' % (c.get('encode_GBps'), c.get('decode_GBps'), c.get('compression_ratio')))" >> $out
done
cat $out
