#!/bin/bash
# k_zplan: the Huffman weights' bit reader on the staged LDS copy instead of global memory
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
O=gpurun_out/r05_zplan_lds_reader.txt; : > $O
python3 -m pytest tests/test_gpu_zstd.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -2 >> $O
for d in narrow int4 wide; do
echo "== phases, $d, 12288 blocks" >> $O
CRYO_CODEC_LIB=$PWD/profiles/variants_zprof.so CRYO_ZSTD_STATS=1 python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --workload zstd_decode --dist $d --blocks 12288 2>&1 | grep "k_zplan:" | head -1 | cut -c1-300 >> $O
done
for shape in "--dist narrow" "--dist int4" "" "--block-size 1048576 --blocks 8192 --dist narrow" "--blocks 1" "--block-size 1048576 --blocks 1"; do
  echo "== zstd_decode $shape" >> $O
  python3 profiles/scripts/ab.py --steps 10 --reps 3 --args "--workload zstd_decode $shape" prod >> $O 2>&1
done
python3 profiles/scripts/ab.py --prof --steps 5 --args "--workload zstd_decode --dist narrow" prod >> $O 2>&1
cat $O
