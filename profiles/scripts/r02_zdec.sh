#!/bin/bash
# zstd decode: tests, bench, kernel stats
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
[ -n "$SKIP_TESTS" ] || timeout 1200 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --workload zstd_decode --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600
rm -rf gpurun_out/zdec_prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/zdec_prof -o zd -- python3 bench.py --workload zstd_decode --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/zdec_prof/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.5: print('%-40s calls %5s avg %9.3f ms %5s%%'%(r['Name'].split('(')[0][-40:],r['Calls'],float(r['AverageNs'])/1e6,r['Percentage']))
PY
