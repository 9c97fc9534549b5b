#!/bin/bash
# zstd encode rate by level (128 KiB and 1 MiB wide blocks) + the affected tests
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r02_zstd_levels.txt; : > $out
for lvl in 5 6 7 8 9 10; do
  timeout 300 python bench.py --workload zstd --level $lvl --blocks 4096 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('level $lvl 128KiB x4096: encode %s GB/s decode %s GB/s ratio %s' % (c.get('encode_GBps'), c.get('decode_GBps'), c.get('compression_ratio')))" >> $out
done
for lvl in 7 10; do
  timeout 300 python bench.py --workload zstd --level $lvl --blocks 512 --block-size 1048576 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('level $lvl 1MiB x512: encode %s GB/s decode %s GB/s ratio %s' % (c.get('encode_GBps'), c.get('decode_GBps'), c.get('compression_ratio')))" >> $out
done
cat $out
timeout 1500 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_host.py -x -q -m gpu 2>&1 | tail -8 | tee -a $out
