#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for pad in ${PADS:-0 256 4352}; do
rm -rf /tmp/abk; CRYO_BENCH_OUT_PAD=$pad timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o r -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline ${WL:-} > /tmp/o.log 2>&1
echo "out pad $pad: $(python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/abk/**/*kernel_stats.csv',recursive=True)[0]
print(' | '.join('%s %.3f'%(r['Name'].split('(')[0].split('::')[-1][:14],float(r['AverageNs'])/1e6) for r in csv.DictReader(open(f)) if ('k_lz4' in r['Name'] or 'k_z' in r['Name']) and 'enc' not in r['Name'] and float(r['AverageNs'])>1e5))
PY
) $(tail -1 /tmp/o.log | cut -c1-0)"
done
