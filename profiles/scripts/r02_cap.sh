#!/bin/bash
# index row stride sweep: cap = B/8 + 64 + pad entries of 2 bytes (32 896 B + 2 pad)
cd "$GRAFT_REPO_ROOT"
for pad in ${PADS:-0 64 960}; do
echo "cap pad $pad (row stride $((34816 + 2*pad)) B): $(for i in 1 2 3; do CRYO_LZ4_IDX_CAP_PAD=$pad timeout 120 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], end=' ')"; done)"
done
