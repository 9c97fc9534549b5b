#!/bin/bash
# zstd encode at the btlazy2 levels (bit-exactness is the GPU suite's; this is the rate)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r03_zstd_btlazy2.txt; : > $out
for spec in "11 131072 2048" "12 131072 2048" "13 1048576 256" "15 1048576 256" "10 16384 8192"; do
  set -- $spec
  timeout 600 python3 bench.py --workload zstd --level $1 --blocks $3 --block-size $2 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('level $1, $3 x $2 bytes: encode %s GB/s, decode %s GB/s, ratio %s' % (c.get('encode_GBps'), c.get('decode_GBps'), c.get('compression_ratio')))" >> $out
done
cat $out
