#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_ab7
bz() { timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python3 -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('$*', '| GB/s', d['value'], 'ms', d['ms_per_step'])
except Exception as e:
    print('$*', '| FAILED', l[-300:])
"; }
{
for v in orig A B; do CRYO_CODEC_LIB=$(pwd)/profiles/variants_zlb_$v.so bz "LaneBits $v"; done
bz "LaneBits all three (production)"
for v in orig A B; do CRYO_CODEC_LIB=$(pwd)/profiles/variants_zlb_$v.so bz "LaneBits $v again"; done
} 2>&1 | tee gpurun_out/r03_ab7/out.txt
