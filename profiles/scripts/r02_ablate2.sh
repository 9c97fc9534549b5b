#!/bin/bash
# what each part of the copy engine costs in kernel time (stats build; the ablated runs produce wrong bytes)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for a in 0 8 16 32 56 2 1; do
rm -rf /tmp/abl; CRYO_LZ4_ABLATE=$a CRYO_LZ4_STATS=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o r -- python3 bench.py --no-cpu-baseline --no-verify --steps 3 --warmup 1 > /dev/null 2>&1
echo "ablate $a: $(python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/abl/**/*kernel_stats.csv',recursive=True)[0]
print(' | '.join('%s %.3f'%(r['Name'].split('(')[0].split('::')[-1][:22],float(r['AverageNs'])/1e6) for r in csv.DictReader(open(f)) if 'dec_seq' in r['Name']))
PY
)"
done
