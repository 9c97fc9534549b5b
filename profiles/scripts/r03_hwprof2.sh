#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_hwprof
timeout 1500 python3 -m pytest tests/test_gpu_zstd.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
export CRYO_CODEC_LIB=$(pwd)/profiles/variants_hwprof.so CRYO_ZSTD_STATS=1 CRYO_ZSTD_LANES=1
timeout 300 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep "zstd pipe" | grep -v "tile" | head -2 | tee gpurun_out/r03_hwprof/log2.txt
