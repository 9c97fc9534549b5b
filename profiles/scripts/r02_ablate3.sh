#!/bin/bash
# what each part of the copy engine costs in kernel time: production kernel with parts compiled out (wrong bytes)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
# build the variants first: for a in 8 16 32 56; do rm -f lz4_dec2.o; make -C pg_cryogen_amd/csrc EXTRA=-DCRYO_ABL=$a; cp pg_cryogen_amd/libcryo_codec.so pg_cryogen_amd/libcryo_codec_abl$a.so; done; then rebuild the product
# (round 3: the variants are loaded through CRYO_CODEC_LIB -- pg_cryogen_amd/_loader.py -- instead of being copied over the
# product library, which an interrupted run left in place)
for a in ${ABLS:-0 8 16 32 56 0}; do
[ $a = 0 ] && unset CRYO_CODEC_LIB || export CRYO_CODEC_LIB=$(pwd)/pg_cryogen_amd/libcryo_codec_abl$a.so
rm -rf /tmp/abl; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o r -- python3 bench.py --no-cpu-baseline --no-verify --steps 10 --warmup 2 > /dev/null 2>&1
echo "ablate $a: $(python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/abl/**/*kernel_stats.csv',recursive=True)[0]
print(' | '.join('%s %.3f'%(r['Name'].split('(')[0].split('::')[-1][:22],float(r['AverageNs'])/1e6) for r in csv.DictReader(open(f)) if 'dec_seq' in r['Name']))
PY
)"
done
