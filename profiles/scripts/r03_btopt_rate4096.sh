#!/bin/bash
# the deep levels with the whole grid in flight (16 workgroups per CU = 4096 waves)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r03_zstd_btopt_rate4096.txt; : > $out
for spec in "11 131072 4096" "13 131072 4096" "16 131072 4096" "19 131072 4096" "22 131072 4096"; do
  set -- $spec
  timeout 900 python3 bench.py --workload zstd --level $1 --blocks $3 --block-size $2 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('level $1, $3 x $2 bytes: encode %s GB/s, ratio %s' % (c.get('encode_GBps'), c.get('compression_ratio')))" >> $out
done
cat $out
