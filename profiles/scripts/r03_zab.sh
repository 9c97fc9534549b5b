#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python3 -m pytest tests/test_gpu_zstd.py -x -q -m gpu 2>&1 | tail -2
b() { timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 8 --warmup 2 "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for v in prev cur prev cur; do
if [ $v = cur ]; then unset CRYO_CODEC_LIB; else export CRYO_CODEC_LIB=$(pwd)/profiles/variants_$v.so; fi
echo "$v: 65536 x 128 KiB: $(b) | 1 MiB x 8192: $(b --block-size 1048576 --blocks 8192)"
done
