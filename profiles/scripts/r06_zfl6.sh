#!/bin/bash
# round 6: the fused kernel with tagged entries: phase split (CRYO_ZSTD_STATS), HBM-side traffic, per-distribution rates
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_zfl6; mkdir -p $O
bash profiles/scripts/build_variant.sh dbg "-DCRYO_DEBUG" zstd_enc.hip > $O/build.txt 2>&1
CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZSTD_STATS=1 timeout 600 python bench.py --workload zstd --steps 1 --warmup 0 --no-cpu-baseline > $O/stats.txt 2> $O/stats.err
grep "zstd enc" $O/stats.err
for d in wide narrow int4 random zeros; do
  echo "== $d" >> $O/ab.txt
  timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline --dist $d >> $O/ab.txt 2>> $O/ab.err
done
echo "== wide_1MiB" >> $O/ab.txt; timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline --block-size 1048576 --blocks 8192 >> $O/ab.txt 2>> $O/ab.err
for l in -5 -1 2 3; do echo "== level$l" >> $O/ab.txt; timeout 600 python bench.py --workload zstd --level $l --steps 2 --warmup 1 --no-cpu-baseline >> $O/ab.txt 2>> $O/ab.err; done
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/fetch -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload zstd --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/write -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload zstd --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/write.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, csv, glob, collections
name=None
for l in open('gpurun_out/r06_zfl6/ab.txt'):
    if l.startswith('=='): name=l.strip(); continue
    try: j=json.loads(l)
    except Exception: continue
    c=j.get('config',{})
    print(name, j.get('value'), {k:v for k,v in c.items() if 'GBps' in k or 'ratio' in k})
for w in ('fetch','write'):
    acc=collections.defaultdict(float)
    for f in glob.glob('gpurun_out/r06_zfl6/%s/**/*counter_collection.csv'%w, recursive=True):
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name']
            k='k_zstd_enc' if 'k_zstd_enc' in n else n.split('(')[0][-30:]
            acc[k]+=float(r['Counter_Value'])
    print(w, {k:'%.4g'%v for k,v in acc.items()})
PY
