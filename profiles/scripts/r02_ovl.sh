#!/bin/bash
cd "$GRAFT_REPO_ROOT"
CRYO_LZ4_INDEX_MIN=1 timeout 900 python -m pytest tests/test_gpu_lz4.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['ms_per_step'])"
bash profiles/scripts/r02_stats.sh
