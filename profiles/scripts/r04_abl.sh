#!/bin/bash
# round 4: what each part of k_lz4_dec_seq costs in kernel time -- production kernel with parts compiled out (wrong bytes, --no-verify):
# CRYO_ABL bits: 8 literal lane runs, 16 independent-match lane runs, 32 match space, 64 flush (output stores), 128 bitmap + chunk bases
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_abl
{
python3 profiles/scripts/ab.py --prof --args=--no-verify --steps 10 prod abl8 abl16 abl24 abl32 abl56 abl64 abl128 abl184 abl248 prod
} 2>&1 | tee gpurun_out/r04_abl/out.txt
