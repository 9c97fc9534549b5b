#!/bin/bash
# the index pass is bimodal between processes (3.05 / 3.57 ms): sample it
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4 5 6 7 8; do
timeout 300 python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
