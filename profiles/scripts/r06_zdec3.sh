#!/bin/bash
# round 6: k_zhufw without its LDS ring (walkers read a sliding window in registers: 12.6 KiB of LDS per wave instead of 22.5)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_zdec3; mkdir -p $O
V=${1:-zhw}
CRYO_CODEC_LIB=profiles/variants_$V.so timeout 1500 python -m pytest tests/test_gpu_zstd.py -x -q 2>&1 | tail -3
row() { local label=$1; shift; env "$@" | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-44s %8.1f GB/s  %8.3f ms' % ('$label', d['value'], r['avg_launch_ms']))"; }
for v in "" $V; do
  L=""; [ -n "$v" ] && L="CRYO_CODEC_LIB=profiles/variants_$v.so"
  row "${v:-production} wide" X=1 $L python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null
  row "${v:-production} narrow" X=1 $L python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline --dist narrow 2>/dev/null
  row "${v:-production} level 5 16384" X=1 $L python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline --level 5 --blocks 16384 2>/dev/null
  row "${v:-production} 8192 x 1 MiB" X=1 $L python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline --block-size 1048576 --blocks 8192 2>/dev/null
  row "${v:-production} 1024 x 128 KiB" X=1 $L python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline --blocks 1024 2>/dev/null
done
