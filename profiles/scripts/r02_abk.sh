#!/bin/bash
# A/B of two builds, per-kernel times by rocprofv3
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
prof() { rm -rf /tmp/abk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o r -- python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-3} --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/abk/**/*kernel_stats.csv',recursive=True)[0]
print(' | '.join('%s %.3f ms'%(r['Name'].split('(')[0][-22:],float(r['AverageNs'])/1e6) for r in csv.DictReader(open(f)) if 'lz4_index' in r['Name'] or 'dec_seq' in r['Name'] or 'dec_ring' in r['Name']))
PY
}
# (round 3: build A is loaded through CRYO_CODEC_LIB instead of being copied over the product library)
echo "B: $(prof "$@")"
echo "A: $(CRYO_CODEC_LIB=$(pwd)/pg_cryogen_amd/libcryo_codec_A.so prof "$@")"
echo "B: $(prof "$@")"
echo "A: $(CRYO_CODEC_LIB=$(pwd)/pg_cryogen_amd/libcryo_codec_A.so prof "$@")"
