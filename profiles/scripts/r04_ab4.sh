#!/bin/bash
# round 4, fourth A/B: copy engine v2 (lane-run reads before writes; match space four chunks at a time) = prod, against v1
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_ab4
O=gpurun_out/r04_ab4
{
timeout 900 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_zstd.py -x -q 2>&1 | tail -3
python3 profiles/scripts/ab.py --prof copyv1 prod
python3 profiles/scripts/ab.py r03base copyv1 prod
for d in narrow int4 random zeros; do python3 profiles/scripts/ab.py --args "--dist $d" copyv1 prod; done
python3 profiles/scripts/ab.py --args "--workload zstd_decode" --steps 6 copyv1 prod
export CRYO_CODEC_LIB=$(pwd)/profiles/variants_debug.so CRYO_LZ4_STATS=1
timeout 600 python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 2>&1 | grep "lz4 seq" | tail -11
} 2>&1 | tee $O/out.txt
