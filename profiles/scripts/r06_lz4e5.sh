#!/bin/bash
# round 6: LZ4 encoder with the backward extension and the emission off the chain (sequences queued, written 64 at a time)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_lz4e5; mkdir -p $O; rm -f $O/ab.txt $O/ab.err
export CRYO_CODEC_LIB=profiles/variants_defer.so
timeout 1200 python -m pytest tests/test_gpu_lz4.py -x -q 2>&1 | tail -3
for d in wide narrow int4 random zeros; do
  echo "== defer $d" >> $O/ab.txt
  timeout 600 python bench.py --workload lz4 --steps 3 --warmup 1 --no-cpu-baseline --dist $d >> $O/ab.txt 2>> $O/ab.err
done
echo "== defer wide_1MiB" >> $O/ab.txt
timeout 600 python bench.py --workload lz4 --steps 3 --warmup 1 --no-cpu-baseline --block-size 1048576 --blocks 8192 >> $O/ab.txt 2>> $O/ab.err
echo "== defer narrow_1MiB" >> $O/ab.txt
timeout 600 python bench.py --workload lz4 --steps 3 --warmup 1 --no-cpu-baseline --block-size 1048576 --blocks 8192 --dist narrow >> $O/ab.txt 2>> $O/ab.err
CRYO_CODEC_LIB=profiles/variants_deferp.so timeout 600 python bench.py --workload lz4 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/prof.txt
grep "lz4 enc" $O/prof.txt | tail -2
grep -v "amdgpu.ids" $O/ab.err | head
python - <<'PY'
import json
name=None
for l in open('gpurun_out/r06_lz4e5/ab.txt'):
    if l.startswith('=='): name=l.strip(); continue
    try: j=json.loads(l)
    except Exception: continue
    c=j.get('config',{})
    print(name, c.get('encode_GBps'))
PY
