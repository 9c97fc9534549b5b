#!/bin/bash
# index kernel ring size experiment
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
CRYO_LZ4_IDX_RING=256 CRYO_LZ4_INDEX_MIN=1 timeout 900 python -m pytest tests/test_gpu_lz4.py -x -q -m gpu 2>&1 | tail -3
for nb in 65536 131072; do for ring in 512 256; do
echo "blocks $nb ring $ring: $(CRYO_LZ4_IDX_RING=$ring timeout 600 python bench.py --blocks $nb --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['ms_per_step'])")"
done; done
