#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_zdec2; mkdir -p $O
bash profiles/scripts/build_variant.sh occ5 "-DCRYO_ZEXEC_OCC=5" zstd_pipe.hip > $O/build.txt 2>&1
bash profiles/scripts/build_variant.sh occ4 "-DCRYO_ZEXEC_OCC=4" zstd_pipe.hip >> $O/build.txt 2>&1
row() { local label=$1; shift; env "$@" | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-44s %8.1f GB/s  %8.3f ms' % ('$label', d['value'], r['avg_launch_ms']))"; }
for v in "" occ5 occ4; do
  L=""; [ -n "$v" ] && L="CRYO_CODEC_LIB=profiles/variants_$v.so"
  row "occ ${v:-6 (production)} wide" X=1 $L python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null
  row "occ ${v:-6} level 5 16384" X=1 $L python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline --level 5 --blocks 16384 2>/dev/null
  row "occ ${v:-6} 8192 x 1 MiB" X=1 $L python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline --block-size 1048576 --blocks 8192 2>/dev/null
done
