#!/bin/bash
# zstd decode: k_zchain + k_zmat (serial chain / parallel values) against k_zseq + k_zrep; parity first
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_zchain
O=gpurun_out/r03_zchain
export TMPDIR=/tmp
bz() { timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python3 -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('$*', '| GB/s', d['value'], 'ms', d['ms_per_step'])
except Exception as e:
    print('$*', '| FAILED', l[-300:])
"; }
{
timeout 1500 python3 -m pytest tests/test_gpu_zstd.py -x -q -m gpu 2>&1 | tail -5
CRYO_ZSTD_STATS=1 timeout 300 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep "zstd pipe" | head -3
bz "zchain+zmat"
CRYO_ZHUF_OLD=1 bz "old zhuf"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o zd -- python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 4 --warmup 1 > $O/prof.log 2>&1
CRYO_ZHUF_OLD=1 CRYO_ZSTD_LANES=1 bz "old zhuf one tile in flight"
CRYO_ZSTD_LANES=1 bz "zchain+zmat one tile in flight"
CRYO_ZSTD_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o zd -- python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 4 --warmup 1 > $O/prof1.log 2>&1
python3 - <<'PY'
import csv, glob
for dd in ('stats', 'stats1'):
  f = glob.glob('gpurun_out/r03_zchain/%s/**/*kernel_stats.csv' % dd, recursive=True)[0]
  print(dd)
  for r in csv.DictReader(open(f)):
    print("%-50s calls %5s avg %10.3f ms %6s%%" % (r['Name'].split('(')[0][-50:], r['Calls'], float(r['AverageNs'])/1e6, r['Percentage']))
PY
} 2>&1 | tee $O/log.txt
