#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_ab3
bl() { timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 2>&1 | tail -1 | python3 -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('$*', '| GB/s', d['value'], 'ms', d['ms_per_step'])
except Exception as e:
    print('$*', '| FAILED', l[-300:])
"; }
{
bl "production (index latency 1)"
CRYO_CODEC_LIB=$(pwd)/profiles/variants_ix_trash.so bl "index unconditional commit stores"
bl "production again"
CRYO_CODEC_LIB=$(pwd)/profiles/variants_ix_trash.so bl "index unconditional commit stores again"
timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | cut -c1-400
timeout 900 python3 -m pytest tests/test_gpu_zstd.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -3
} 2>&1 | tee gpurun_out/r03_ab3/out.txt
