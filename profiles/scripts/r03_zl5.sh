#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_zl5
O=gpurun_out/r03_zl5
export TMPDIR=/tmp
CRYO_ZSTD_STATS=1 timeout 300 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 1 --warmup 0 --blocks 16384 --level 5 2>&1 | grep "zstd pipe" | cut -c1-400 | head -3
CRYO_ZSTD_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o zd -- python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 4 --warmup 1 --blocks 16384 --level 5 > $O/prof1.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r03_zl5/stats1/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_z' in r['Name'] and 'enc' not in r['Name']:
      print("  %-30s calls %5s avg %10.3f ms" % (r['Name'].split('(')[0][-30:], r['Calls'], float(r['AverageNs'])/1e6))
PY
