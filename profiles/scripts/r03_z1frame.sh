#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_z1f
O=gpurun_out/r03_z1f
export TMPDIR=/tmp
for spec in "1048576 1" "131072 1" "1048576 16"; do
set -- $spec
rm -rf $O/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o zd -- python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 10 --warmup 2 --block-size $1 --blocks $2 > $O/prof.log 2>&1
echo "== $2 x $1"
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r03_z1f/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_z' in r['Name'] and 'enc' not in r['Name']:
      print("  %-30s calls %5s avg %10.3f ms" % (r['Name'].split('(')[0][-30:], r['Calls'], float(r['AverageNs'])/1e6))
PY
done
