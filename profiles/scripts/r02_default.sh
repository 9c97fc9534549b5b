#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4; do timeout 200 python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['steps'], d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
timeout 300 python3 bench.py --gpus 2 --steps 40 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2 ranks', d['ms_per_step'], d['value'])"
timeout 300 python3 bench.py --blocks 131072 --steps 40 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('131072 blocks', d['ms_per_step'], d['value'], d['roofline']['frac'])"
