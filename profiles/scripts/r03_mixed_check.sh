#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_mixed
bm() { timeout 600 python3 bench.py --workload mixed --steps 20 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', '| GB/s', d['value'], 'ms', d['ms_per_step'])"; }
bl() { timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', '| GB/s', d['value'], 'ms', d['ms_per_step'], 'ratio', d['config']['compression_ratio'])"; }
{
bm "mixed default"
CRYO_LZ4_DECODE_PATH=1 bm "mixed, lz4 half on the in-wave parser"
CRYO_ZSTD_LANES=1 bm "mixed, one zstd tile in flight"
CRYO_ZSTD_DECODE_PATH=1 bm "mixed, zstd fused"
bl --blocks 8192 --accel 50
bl --blocks 8192 --accel 50 --lz4-path 1
bl --blocks 8192 --accel 50 --lz4-walkers 1
bl --blocks 8192 --accel 50 --lz4-walkers 2
bl --blocks 8192 --accel 50 --lz4-walkers 4
bl --blocks 65536 --accel 50
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_mixed/stats -o run -- python3 $R/bench.py --workload mixed --steps 10 --warmup 2 > $R/gpurun_out/r03_mixed/stats.log 2>&1
cd $R
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03_mixed/stats/run_kernel_stats.csv')):
    print("%-44s calls %4s avg %10.1f us" % (r['Name'].split('(')[0][-44:], r['Calls'], float(r['AverageNs'])/1e3))
PY
} 2>&1 | tee gpurun_out/r03_mixed/out.txt
