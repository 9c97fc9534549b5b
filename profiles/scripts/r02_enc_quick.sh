#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for wl in zstd lz4; do
timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$wl: value %s encode %s GB/s decode %s GB/s' % (d['value'], c.get('encode_GBps'), c.get('decode_GBps')))"
done
