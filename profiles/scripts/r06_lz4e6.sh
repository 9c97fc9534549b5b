#!/bin/bash
# round 6: LZ4 encoder after the queue / early far loads / 12-byte compares: tags and ring size again
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_lz4e6; mkdir -p $O; rm -f $O/ab.txt $O/ab.err
export CRYO_CODEC_LIB=profiles/variants_defer.so
for cfg in "0 1" "0 2" "1 1" "1 2"; do set -- $cfg
  for d in wide narrow; do
    echo "== tags$1 ring${2}k $d" >> $O/ab.txt
    CRYO_LZ4_ENC_TAGS=$1 CRYO_LZ4_ENC_WINDOW=$2 timeout 600 python bench.py --workload lz4 --steps 3 --warmup 1 --no-cpu-baseline --dist $d >> $O/ab.txt 2>> $O/ab.err
  done
done
python - <<'PY'
import json
name=None
for l in open('gpurun_out/r06_lz4e6/ab.txt'):
    if l.startswith('=='): name=l.strip(); continue
    try: j=json.loads(l)
    except Exception: continue
    print(name, j.get('config',{}).get('encode_GBps'))
PY
