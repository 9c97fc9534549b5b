#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_z2
{
for lv in 1 3 5; do
echo "== level $lv"
CRYO_CODEC_LIB=$(pwd)/profiles/variants_zdebug.so CRYO_ZSTD_STATS=1 timeout 600 python3 bench.py --workload zstd_decode --level $lv --blocks 8192 --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep "zstd pipe" | head -4
done
echo "== 1 MiB level 1"
CRYO_CODEC_LIB=$(pwd)/profiles/variants_zdebug.so CRYO_ZSTD_STATS=1 timeout 600 python3 bench.py --workload zstd_decode --blocks 1024 --block-size 1048576 --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep "zstd pipe" | head -2
} 2>&1 | tee gpurun_out/r04_z2/out.txt
