#!/bin/bash
# per-kernel times of the zstd decode pipeline, one and two tiles in flight
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_zstats
O=gpurun_out/r03_zstats
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o zd -- python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 4 --warmup 1 > $O/prof.log 2>&1
CRYO_ZSTD_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o zd -- python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 4 --warmup 1 > $O/prof1.log 2>&1
python3 - <<'PY' | tee $O/log.txt
import csv, glob, json
for dd in ('stats', 'stats1'):
  f = glob.glob('gpurun_out/r03_zstats/%s/**/*kernel_stats.csv' % dd, recursive=True)[0]
  print(dd, open('gpurun_out/r03_zstats/prof%s.log' % ('1' if dd == 'stats1' else '')).read().strip().split('\n')[-1][:60])
  for r in csv.DictReader(open(f)):
    if 'k_z' in r['Name'] and 'enc' not in r['Name']:
      print("  %-30s calls %5s avg %10.3f ms" % (r['Name'].split('(')[0][-30:], r['Calls'], float(r['AverageNs'])/1e6))
PY
