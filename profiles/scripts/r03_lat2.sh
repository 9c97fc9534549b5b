#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_lat2
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_lz4.py -x -q -m gpu -k "few_blocks" 2>&1 | grep -E "Error|error|assert|passed|failed" | head -12
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_lat2/stats -o lat -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --block-size 1048576 --blocks 16 > gpurun_out/r03_lat2/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r03_lat2/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'lat' in r['Name'] or 'lz4' in r['Name']:
        print("  %-36s calls %5s avg %10.3f ms total %10.3f ms" % (r['Name'].split('(')[0][-36:], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
PY
