#!/bin/bash
# k_zhufw: walkers per stream for short streams (T >> seglog), and where the wave time goes on `narrow` (CRYO_HW_PROF build)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
O=gpurun_out/r05_hufw_seglog.txt; : > $O
for shape in "--dist narrow" "--dist int4" "--blocks 16384" "--block-size 1048576 --blocks 8192 --dist narrow" "--blocks 16384 --level -5"; do
  for sl in 11 10 9 8; do
    echo "== workload zstd_decode $shape : CRYO_ZHUFW_SEGLOG=$sl" >> $O
    python3 profiles/scripts/ab.py --steps 10 --reps 2 --env CRYO_ZHUFW_SEGLOG=$sl --args "--workload zstd_decode $shape" zdbg >> $O 2>&1
  done
done
for sl in 11 10 9; do
echo "== phases, narrow, 12288 blocks, seglog $sl" >> $O
CRYO_CODEC_LIB=$PWD/profiles/variants_zprof.so CRYO_ZSTD_STATS=1 CRYO_ZHUFW_SEGLOG=$sl python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --workload zstd_decode --dist narrow --blocks 12288 2>&1 | grep "zstd pipe" | head -8 | cut -c1-600 >> $O
done
python3 profiles/scripts/ab.py --prof --steps 5 --env CRYO_ZHUFW_SEGLOG=10 --args "--workload zstd_decode --dist narrow" zdbg >> $O 2>&1
cat $O
