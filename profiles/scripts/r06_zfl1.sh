#!/bin/bash
# round 6, first GPU contact of the two-kernel `fast` encoder (k_zfind + k_zent): correctness, then A/B rows
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_zfl1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zstd.py -x -q -k "encode or roundtrip or corners" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
bash profiles/scripts/build_variant.sh dbg "-DCRYO_DEBUG" zstd_enc.hip > $O/build.txt 2>&1
bash profiles/scripts/build_variant.sh dbgnorev "-DCRYO_DEBUG -DCRYO_ZFL_REV=0" zstd_enc.hip >> $O/build.txt 2>&1
run() { # name, env...
  local name=$1; shift
  echo "== $name" >> $O/ab.txt
  env "$@" timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline >> $O/ab.txt 2>> $O/ab.err
}
run old        CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZFL=0
run lpb64      CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZFL_LPB=64 CRYO_ZFL_STATS=1
run lpb32      CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZFL_LPB=32 CRYO_ZFL_STATS=1
run lpb64norev CRYO_CODEC_LIB=profiles/variants_dbgnorev.so CRYO_ZFL_LPB=64 CRYO_ZFL_STATS=1
run lpb32norev CRYO_CODEC_LIB=profiles/variants_dbgnorev.so CRYO_ZFL_LPB=32 CRYO_ZFL_STATS=1
for d in narrow int4 zeros random; do
  echo "== prod_$d" >> $O/ab.txt
  timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline --dist $d >> $O/ab.txt 2>> $O/ab.err
done
grep -h "zfind\|Error\|error\|assert" $O/ab.err | head -40
python - <<'PY'
import json,re
name=None
for l in open('gpurun_out/r06_zfl1/ab.txt'):
    if l.startswith('=='): name=l.strip(); continue
    try: j=json.loads(l)
    except Exception: continue
    c=j.get('config',{})
    print(name, j.get('value'), {k:v for k,v in c.items() if 'GBps' in k or 'ratio' in k})
PY
