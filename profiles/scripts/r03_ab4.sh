#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_ab4
bl() { timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 2>&1 | tail -1 | python3 -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('$*', '| GB/s', d['value'], 'ms', d['ms_per_step'])
except Exception as e:
    print('$*', '| FAILED', l[-300:])
"; }
{
CRYO_CODEC_LIB=$(pwd)/profiles/variants_ix_v2.so timeout 900 python3 -m pytest tests/test_gpu_lz4.py -x -q -m gpu 2>&1 | tail -3
CRYO_CODEC_LIB=$(pwd)/profiles/variants_ix_v2.so bl "index turn v2"
CRYO_CODEC_LIB=$(pwd)/profiles/variants_ix_trash.so bl "index v1 + unconditional commit"
CRYO_CODEC_LIB=$(pwd)/profiles/variants_ix_v2.so bl "index turn v2 again"
CRYO_CODEC_LIB=$(pwd)/profiles/variants_ix_trash.so bl "index v1 + unconditional commit again"
} 2>&1 | tee gpurun_out/r03_ab4/out.txt
