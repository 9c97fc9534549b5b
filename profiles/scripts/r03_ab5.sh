#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_ab5
bl() { timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 "$@" 2>&1 | tail -1 | python3 -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('$*', '| GB/s', d['value'], 'ms', d['ms_per_step'])
except Exception as e:
    print('$*', '| FAILED', l[-300:])
"; }
{
timeout 900 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_zstd.py -x -q -m gpu 2>&1 | tail -3
bl
CRYO_CODEC_LIB=$(pwd)/profiles/variants_ix_v2.so bl
bl
CRYO_CODEC_LIB=$(pwd)/profiles/variants_ix_v2.so bl
bl --blocks 16384
bl --blocks 4096
bl --block-size 1048576 --blocks 8192
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_ab5/stats -o run -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 2 > $R/gpurun_out/r03_ab5/stats.log 2>&1
cd $R
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03_ab5/stats/run_kernel_stats.csv')):
    print("%-44s calls %4s avg %10.1f us" % (r['Name'].split('(')[0][-44:], r['Calls'], float(r['AverageNs'])/1e3))
PY
} 2>&1 | tee gpurun_out/r03_ab5/out.txt
