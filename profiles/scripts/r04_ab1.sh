#!/bin/bash
# round 4, first A/B: index pass with hand-placed vmcnt waits (1 / 2 rounds of distance) against round 3's; decoder ring 8 KiB
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_ab1
O=gpurun_out/r04_ab1
{
timeout 900 python3 -m pytest tests/test_gpu_lz4.py -x -q 2>&1 | tail -3
python3 profiles/scripts/ab.py --prof r03base prod idx_d1 dec_r8k_w1 dec_r8k_w3 dec_r8k_w5
python3 profiles/scripts/ab.py r03base prod idx_d1
export CRYO_CODEC_LIB=$(pwd)/profiles/variants_debug.so CRYO_LZ4_STATS=1
timeout 600 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep "lz4 seq" | tail -14
} 2>&1 | tee $O/out.txt
