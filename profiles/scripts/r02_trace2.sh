#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4 5 6 7 8; do
CRYO_BENCH_TRACE=1 timeout 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2> /tmp/tr.err | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
grep "device pointers" /tmp/tr.err
done
