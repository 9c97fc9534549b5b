#!/bin/bash
# zstd decode: parity, two bench runs, per-kernel times with one tile in flight
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_zquick
O=gpurun_out/r03_zquick
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_zstd.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
CRYO_ZSTD_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o zd -- python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 4 --warmup 1 > $O/prof1.log 2>&1
python3 - <<'PY' | tee $O/log.txt
import csv, glob
f = glob.glob('gpurun_out/r03_zquick/stats1/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_z' in r['Name'] and 'enc' not in r['Name']:
      print("  %-30s calls %5s avg %10.3f ms" % (r['Name'].split('(')[0][-30:], r['Calls'], float(r['AverageNs'])/1e6))
PY
