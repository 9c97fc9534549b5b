#!/bin/bash
# round 6: v3 finder (tagged 32-bit table entries): correctness, stats, kernel times
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_zfl4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zstd.py -x -q -k "encode or roundtrip or corners" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
bash profiles/scripts/build_variant.sh dbg "-DCRYO_DEBUG" zstd_enc.hip > $O/build.txt 2>&1
run() { local name=$1; shift; echo "== $name" >> $O/ab.txt; env "$@" timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline >> $O/ab.txt 2>> $O/ab.err; }
run prod X=1
run stats CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZFL_STATS=1
for d in narrow int4 zeros random; do
  echo "== prod_$d" >> $O/ab.txt
  timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline --dist $d >> $O/ab.txt 2>> $O/ab.err
done
grep -h "zfind" $O/ab.err | sort | uniq -c
grep -v zfind $O/ab.err | head -20
bash profiles/quick_stats.sh zstd --steps 2 2>&1 | grep -v "k_z[a-z]*[0-9]* \|synth\|compare\|rocclr" | tail -5
python - <<'PY'
import json
name=None
for l in open('gpurun_out/r06_zfl4/ab.txt'):
    if l.startswith('=='): name=l.strip(); continue
    try: j=json.loads(l)
    except Exception: continue
    c=j.get('config',{})
    print(name, j.get('value'), {k:v for k,v in c.items() if 'GBps' in k})
PY
