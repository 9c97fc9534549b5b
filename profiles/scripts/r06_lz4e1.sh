#!/bin/bash
# round 6: LZ4 encoder with a tag plane and one far trip per sequence
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_lz4e1; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_lz4.py -x -q -k "encode or golden_cells_on_gpu or checksum or single_block" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
for d in wide narrow int4 zeros random; do
  echo "== $d" >> $O/ab.txt
  timeout 600 python bench.py --workload lz4 --steps 3 --warmup 1 --no-cpu-baseline --dist $d >> $O/ab.txt 2>> $O/ab.err
done
echo "== wide_1MiB" >> $O/ab.txt; timeout 600 python bench.py --workload lz4 --steps 3 --warmup 1 --no-cpu-baseline --block-size 1048576 --blocks 8192 >> $O/ab.txt 2>> $O/ab.err
echo "== narrow_1MiB" >> $O/ab.txt; timeout 600 python bench.py --workload lz4 --steps 3 --warmup 1 --no-cpu-baseline --dist narrow --block-size 1048576 --blocks 8192 >> $O/ab.txt 2>> $O/ab.err
echo "== wide_accel50" >> $O/ab.txt; timeout 600 python bench.py --workload lz4 --accel 50 --steps 3 --warmup 1 --no-cpu-baseline >> $O/ab.txt 2>> $O/ab.err
grep -v "amdgpu.ids" $O/ab.err | head
python - <<'PY'
import json
name=None
for l in open('gpurun_out/r06_lz4e1/ab.txt'):
    if l.startswith('=='): name=l.strip(); continue
    try: j=json.loads(l)
    except Exception: continue
    c=j.get('config',{})
    print(name, j.get('value'), {k:v for k,v in c.items() if 'GBps' in k or 'ratio' in k})
PY
bash profiles/scripts/pmc_sq.sh r06_lz4e1 lz4 2>&1 | grep "k_lz4_enc2" | head -3
