#!/bin/bash
# Huffman blocks with few literals: k_zhuf (lane per stream) instead of k_zhufw's walkers?  threshold sweep (debug build).
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
O=gpurun_out/r05_hufw_min.txt; : > $O
for shape in "--dist narrow" "--dist int4" "" "--level 3 --blocks 16384" "--block-size 1048576 --blocks 8192 --dist narrow" "--block-size 1048576 --blocks 8192"; do
  for t in 0 4096 12288 24576 49152 200000; do
    for g in 256 512; do
      [ $t = 0 ] && [ $g = 512 ] && continue
      echo "== workload zstd_decode $shape : CRYO_ZHUFW_MIN=$t grid $g" >> $O
      python3 profiles/scripts/ab.py --steps 10 --env CRYO_ZHUFW_MIN=$t,CRYO_ZHUF2_GRID=$g --args "--workload zstd_decode $shape" zdbg >> $O 2>&1
    done
  done
done
echo "== kernels, narrow, threshold 24576 grid 512" >> $O
python3 profiles/scripts/ab.py --prof --steps 5 --env CRYO_ZHUFW_MIN=24576,CRYO_ZHUF2_GRID=512 --args "--workload zstd_decode --dist narrow" zdbg >> $O 2>&1
python3 profiles/scripts/ab.py --prof --steps 5 --env CRYO_ZHUFW_MIN=0 --args "--workload zstd_decode --dist narrow" zdbg >> $O 2>&1
cat $O
