#!/bin/bash
# zstd decode: fewer streams per wave in the LDS-bound entropy kernels so that plan / exec waves of the other lane's tile co-reside
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_zvariants
O=gpurun_out/r03_zvariants
b() { timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python3 -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('$*', '| GB/s', d['value'], 'ms', d['ms_per_step'])
except Exception as e:
    print('$*', '| FAILED', l[-300:])
"; }
{
for v in 16_29 15_26 14_24 12_22; do
  for lanes in 1 2; do
    export CRYO_CODEC_LIB=$(pwd)/profiles/variants_zp_$v.so CRYO_ZSTD_LANES=$lanes
    b "variant $v lanes $lanes"
  done
done
} 2>&1 | tee $O/sweep.txt
