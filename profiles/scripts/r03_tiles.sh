#!/bin/bash
# (ran at commit 2f6d... of round 3; the tiles option it uses was removed afterwards: the result is profiles/r03_tiles_experiment.txt)
# tiled indexed decode: index pass of tile t+1 on a side stream beside the decoder of tile t
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_tiles
O=gpurun_out/r03_tiles
b() { timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 "$@" 2>&1 | tail -1 | python3 -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('$*', '| GB/s', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
except Exception as e:
    print('$*', '| FAILED', l[-300:])
"; }
{
b --lz4-tiles 1
b --lz4-tiles 2
b --lz4-tiles 4
b --lz4-tiles 8
b --lz4-tiles 16
b --lz4-tiles 4 --lz4-walkers 2
b --lz4-tiles 4 --lz4-walkers 8
b --lz4-tiles 2 --lz4-walkers 4
b --blocks 131072 --lz4-tiles 1
b --blocks 131072 --lz4-tiles 8
b --blocks 32768 --lz4-tiles 4
b --block-size 1048576 --blocks 8192 --lz4-tiles 2
} 2>&1 | tee $O/sweep.txt
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o run -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 --lz4-tiles 4 > $R/$O/stats.log 2>&1
cd $R
python3 - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/r03_tiles/stats/run_kernel_trace.csv')) if 'lz4_index' in r['Kernel_Name'] or 'lz4_dec_seq' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[-40]['Start_Timestamp'])
for r in rows[-40:]:
    print('%-14s start %9.1f us  end %9.1f us  dur %8.1f' % (r['Kernel_Name'].split('(')[0][-14:], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
