#!/bin/bash
# A/B builds: zstd entropy kernels with fewer streams per wave + two tiles in flight; LZ4 index ring size / commit latency
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_variants2
O=gpurun_out/r03_variants2
bz() { timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python3 -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('$*', '| GB/s', d['value'], 'ms', d['ms_per_step'])
except Exception as e:
    print('$*', '| FAILED', l[-300:])
"; }
bl() { timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -1 | python3 -c "
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
    bz "zstd variant $v lanes $lanes"
  done
done
unset CRYO_ZSTD_LANES
for v in 512_1 256_1 256_2; do
  export CRYO_CODEC_LIB=$(pwd)/profiles/variants_ix_$v.so
  bl "lz4 index ring_latency $v"
done
unset CRYO_CODEC_LIB
bl "lz4 production"
} 2>&1 | tee $O/sweep.txt
