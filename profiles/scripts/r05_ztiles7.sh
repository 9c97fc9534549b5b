#!/bin/bash
# equal tiles from 8 193 zstd blocks on (prod) against the rule of one tile up to 12 288 (ztold = tiles of 12 288 throughout)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
O=gpurun_out/r05_zstd_tiles_8193.txt; : > $O
python3 -m pytest tests/test_gpu_zstd.py -m gpu -x -q -k "pipeline_many or threshold" 2>&1 | tail -2 >> $O
for shape in "--blocks 8192" "--blocks 9000" "--blocks 10240" "--blocks 12288" "--blocks 10240 --dist narrow" "--blocks 10240 --level 5" "--block-size 1048576 --blocks 1280" "--block-size 262144 --blocks 5000"; do
  echo "== zstd_decode $shape" >> $O
  python3 profiles/scripts/ab.py --steps 10 --reps 2 --args "--workload zstd_decode $shape" ztold prod >> $O 2>&1
done
cat $O
