#!/bin/bash
# round 6: tagged table entries + two trips per sequence in block_fast_gbatch; k_zfind (global table) + k_zent against the fused kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_zfl5; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_zstd.py -x -q -k "encode or roundtrip or corners" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
bash profiles/scripts/build_variant.sh dbg "-DCRYO_DEBUG" zstd_enc.hip > $O/build.txt 2>&1
run() { local name=$1; shift; echo "== $name" >> $O/ab.txt; env "$@" timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline >> $O/ab.txt 2>> $O/ab.err; }
run split X=1
run fused CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZFL=0
run split_grid4096 CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZFL_GRID=4096
run split_grid3072 CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZFL_GRID=3072
run split_tile16k CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZFL_TILE=16384
for d in narrow int4 random; do
  echo "== split_$d" >> $O/ab.txt
  timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline --dist $d >> $O/ab.txt 2>> $O/ab.err
done
echo "== fused_1MiB" >> $O/ab.txt; timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline --block-size 1048576 --blocks 8192 >> $O/ab.txt 2>> $O/ab.err
grep -v "amdgpu.ids" $O/ab.err | head -20
bash profiles/quick_stats.sh zstd --steps 2 2>&1 | grep "cryo::  \|k_zstd_enc" | tail -5
python - <<'PY'
import json
name=None
for l in open('gpurun_out/r06_zfl5/ab.txt'):
    if l.startswith('=='): name=l.strip(); continue
    try: j=json.loads(l)
    except Exception: continue
    c=j.get('config',{})
    print(name, j.get('value'), {k:v for k,v in c.items() if 'GBps' in k})
PY
