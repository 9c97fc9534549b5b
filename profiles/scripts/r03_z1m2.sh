#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_z1m
O=gpurun_out/r03_z1m
export TMPDIR=/tmp
CRYO_ZSTD_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o zd -- python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 4 --warmup 1 --block-size 1048576 --blocks 8192 > $O/prof1.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r03_z1m/stats1/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_z' in r['Name'] and 'enc' not in r['Name']:
      print("  %-30s calls %5s avg %10.3f ms" % (r['Name'].split('(')[0][-30:], r['Calls'], float(r['AverageNs'])/1e6))
PY
tail -1 $O/prof1.log | cut -c1-200
