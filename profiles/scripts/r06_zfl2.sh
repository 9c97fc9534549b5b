#!/bin/bash
# round 6: per-kernel times of the two-kernel `fast` encoder + phase stamps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_zfl2; mkdir -p $O
bash profiles/scripts/build_variant.sh dbg "-DCRYO_DEBUG" zstd_enc.hip > $O/build.txt 2>&1
run() { local name=$1; shift; echo "== $name" >> $O/ab.txt; env "$@" timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline >> $O/ab.txt 2>> $O/ab.err; }
run prod X=1
run lpb64stats CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZFL_LPB=64 CRYO_ZFL_STATS=1
run lpb32stats CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZFL_LPB=32 CRYO_ZFL_STATS=1
run lpb32 CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZFL_LPB=32
grep -h "zfind" $O/ab.err | sort | uniq -c
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o zstd -- python3 $GRAFT_REPO_ROOT/bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-6 {} | head -12'
python - <<'PY'
import json
name=None
for l in open('gpurun_out/r06_zfl2/ab.txt'):
    if l.startswith('=='): name=l.strip(); continue
    try: j=json.loads(l)
    except Exception: continue
    c=j.get('config',{})
    print(name, j.get('value'), {k:v for k,v in c.items() if 'GBps' in k})
PY
