#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python3 -m pytest tests/test_gpu_zstd.py -x -q -m gpu 2>&1 | tail -2
b() { timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 8 --warmup 2 "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
echo "65536 x 128 KiB: $(b) | 1 MiB x 8192: $(b --block-size 1048576 --blocks 8192)"
export CRYO_CODEC_LIB=$(pwd)/profiles/variants_hwprof.so
CRYO_ZSTD_STATS=1 CRYO_ZSTD_LANES=1 timeout 300 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 1 --warmup 0 --block-size 1048576 --blocks 2048 2>&1 | grep -E "handed back because|k_zhufw wave" | cut -c1-260 | head -2
