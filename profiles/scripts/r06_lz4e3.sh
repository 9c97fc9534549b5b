#!/bin/bash
# round 6: phase stamps of the LZ4 encoder (variant build with -DCRYO_LZ4E_PROF)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_lz4e3; mkdir -p $O
bash profiles/scripts/build_variant.sh prof "-DCRYO_DEBUG -DCRYO_LZ4E_PROF" lz4_enc2.hip > $O/build.txt 2>&1
for d in wide narrow; do
CRYO_CODEC_LIB=profiles/variants_prof.so timeout 600 python bench.py --workload lz4 --steps 1 --warmup 0 --no-cpu-baseline --dist $d 2>&1 | grep "lz4 enc\]" | tail -1
done
CRYO_CODEC_LIB=profiles/variants_prof.so CRYO_LZ4_ENC_DBG=4 timeout 600 python bench.py --workload lz4 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "lz4 enc\]" | tail -1
