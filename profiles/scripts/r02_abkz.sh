#!/bin/bash
# A/B of two builds, per-kernel times of the zstd decode pipeline by rocprofv3
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
prof() { rm -rf /tmp/abk; timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o r -- python3 bench.py --workload zstd_decode --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/abk/**/*kernel_stats.csv',recursive=True)[0]
print(' | '.join('%s %.3f'%(r['Name'].split('(')[0].split('::')[-1],float(r['AverageNs'])/1e6) for r in csv.DictReader(open(f)) if 'k_z' in r['Name'] and 'enc' not in r['Name']))
PY
}
cp pg_cryogen_amd/libcryo_codec.so /tmp/B.so
echo "B: $(prof "$@")"
cp pg_cryogen_amd/libcryo_codec_A.so pg_cryogen_amd/libcryo_codec.so
echo "A: $(prof "$@")"
cp /tmp/B.so pg_cryogen_amd/libcryo_codec.so
