#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for pad in 0 112 4208 65536; do
echo "pad $pad: $(for i in 1 2 3 4 5 6; do CRYO_BENCH_STRIDE_PAD=$pad timeout 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], end=' ')"; done)"
done
