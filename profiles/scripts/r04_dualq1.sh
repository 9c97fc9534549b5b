#!/bin/bash
# two waves per block with a queue between them (k_lz4_dec_dualq): parity first, then lock-step against queue per batch shape
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_dualq1.txt; : > $out
timeout 900 python3 -m pytest tests/test_gpu_lz4.py -x -q -m gpu 2>&1 | tail -6 | tee -a $out
for args in "--blocks 1024" "--blocks 256" "--blocks 1792" "--blocks 64 --lz4-path 2" "--block-size 1048576 --blocks 512" "--block-size 1048576 --blocks 128" "--blocks 1024 --dist narrow" "--blocks 1024 --dist int4"; do
  for wv in 2 3; do
    printf "%-44s waves %d  " "$args" $wv | tee -a $out
    timeout 300 python3 profiles/scripts/ab.py --prof --steps 30 "--args=$args --lz4-waves $wv" prod 2>&1 | tail -1 | sed 's/^prod *//' | tee -a $out
  done
done
