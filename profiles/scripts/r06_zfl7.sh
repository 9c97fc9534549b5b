#!/bin/bash
# round 6: step width / grid of the tagged two-trip finder; long-match counting 16 bytes per lane
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_zfl7; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_zstd.py -x -q -k "encode or roundtrip or corners" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -2 $O/pytest.txt
bash profiles/scripts/build_variant.sh dbg "-DCRYO_DEBUG" zstd_enc.hip > $O/build.txt 2>&1
run() { local name=$1; shift; echo "== $name" >> $O/ab.txt; env "$@" timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline >> $O/ab.txt 2>> $O/ab.err; }
for w in 6 8 12 16 24; do run width$w CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZSTD_ENC_WIDTH=$w; done
for g in 2048 3072; do run grid$g CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZSTD_ENC_GRID=$g; done
for d in narrow int4 zeros; do
  echo "== $d" >> $O/ab.txt
  timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline --dist $d >> $O/ab.txt 2>> $O/ab.err
done
echo "== narrow_1MiB" >> $O/ab.txt; timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline --dist narrow --block-size 1048576 --blocks 8192 >> $O/ab.txt 2>> $O/ab.err
grep -v "amdgpu.ids" $O/ab.err | head
python - <<'PY'
import json
name=None
for l in open('gpurun_out/r06_zfl7/ab.txt'):
    if l.startswith('=='): name=l.strip(); continue
    try: j=json.loads(l)
    except Exception: continue
    c=j.get('config',{})
    print(name, j.get('value'), {k:v for k,v in c.items() if 'GBps' in k})
PY
