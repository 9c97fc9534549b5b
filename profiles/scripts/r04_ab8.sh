#!/bin/bash
# round 4: the decoder's per-wave chain with ONE wave per SIMD (1024 blocks): what each part costs when nothing competes
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_ab8
{
python3 profiles/scripts/ab.py --prof --steps 30 --args="--blocks 1024 --no-verify" prod copyv2 abl8 abl16 abl32 abl56 abl64 abl248
} 2>&1 | tee gpurun_out/r04_ab8/out.txt
