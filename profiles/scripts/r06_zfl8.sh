#!/bin/bash
# round 6: the FSE sequence bitstream on three lanes + packing per sequence
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_zfl8; mkdir -p $O; rm -f $O/ab.txt
timeout 1500 python -m pytest tests/test_gpu_zstd.py -x -q -k "encode or roundtrip or corners or few_seq" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
bash profiles/scripts/build_variant.sh dbg "-DCRYO_DEBUG" zstd_enc.hip > $O/build.txt 2>&1
CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZSTD_STATS=1 timeout 600 python bench.py --workload zstd --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "zstd enc"
for d in wide narrow int4; do
  echo "== $d" >> $O/ab.txt
  timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline --dist $d >> $O/ab.txt 2>> $O/ab.err
done
echo "== wide_1MiB" >> $O/ab.txt; timeout 600 python bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline --block-size 1048576 --blocks 8192 >> $O/ab.txt 2>> $O/ab.err
for l in -5 3 5; do echo "== level$l" >> $O/ab.txt; timeout 600 python bench.py --workload zstd --level $l --steps 2 --warmup 1 --no-cpu-baseline --blocks 16384 >> $O/ab.txt 2>> $O/ab.err; done
grep -v "amdgpu.ids" $O/ab.err | head
python - <<'PY'
import json
name=None
for l in open('gpurun_out/r06_zfl8/ab.txt'):
    if l.startswith('=='): name=l.strip(); continue
    try: j=json.loads(l)
    except Exception: continue
    c=j.get('config',{})
    print(name, c.get('encode_GBps'), c.get('decode_GBps'))
PY
