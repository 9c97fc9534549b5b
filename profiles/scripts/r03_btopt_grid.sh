#!/bin/bash
# level 22 / 13 / 11 encode rate against the number of waves in flight (CRYO_ZSTD_ENC_GRID)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r03_zstd_btopt_grid.txt; : > $out
for lvl in 22 13 11; do
for g in 256 512 1024 2048 2816; do
  CRYO_ZSTD_ENC_GRID=$g timeout 900 python3 bench.py --workload zstd --level $lvl --blocks 2816 --block-size 131072 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('level $lvl, grid $g: encode %s GB/s' % (c.get('encode_GBps'),))" >> $out
done
done
cat $out
