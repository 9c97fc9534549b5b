#!/bin/bash
# round 6: LZ4 encoder: tags (12 workgroups per CU) against packed high bits without tags (16 per CU), 1 KiB / 2 KiB ring
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_lz4e4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lz4.py -x -q -k "encode or golden_cells_on_gpu or checksum or single_block" 2>&1 | tail -2
bash profiles/scripts/build_variant.sh dbg "-DCRYO_DEBUG" lz4_enc2.hip > $O/build.txt 2>&1
for t in 0 1; do for w in 1 2; do
  CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_LZ4_ENC_TAGS=$t CRYO_LZ4_ENC_WINDOW=$w timeout 300 python -m pytest tests/test_gpu_lz4.py -x -q -k "encode or golden_cells_on_gpu" 2>&1 | tail -1
  for d in wide narrow int4 random; do
    echo "== tags$t ring${w}k $d" >> $O/ab.txt
    CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_LZ4_ENC_TAGS=$t CRYO_LZ4_ENC_WINDOW=$w timeout 600 python bench.py --workload lz4 --steps 3 --warmup 1 --no-cpu-baseline --dist $d >> $O/ab.txt 2>> $O/ab.err
  done
  echo "== tags$t ring${w}k wide_1MiB" >> $O/ab.txt
  CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_LZ4_ENC_TAGS=$t CRYO_LZ4_ENC_WINDOW=$w timeout 600 python bench.py --workload lz4 --steps 3 --warmup 1 --no-cpu-baseline --block-size 1048576 --blocks 8192 >> $O/ab.txt 2>> $O/ab.err
done; done
grep -v "amdgpu.ids" $O/ab.err | head
python - <<'PY'
import json
name=None
for l in open('gpurun_out/r06_lz4e4/ab.txt'):
    if l.startswith('=='): name=l.strip(); continue
    try: j=json.loads(l)
    except Exception: continue
    c=j.get('config',{})
    print(name, c.get('encode_GBps'))
PY
