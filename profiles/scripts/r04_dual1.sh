#!/bin/bash
# two waves per block (k_lz4_dec_dual): parity first, then the mid-size batches against one wave per block
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_dual1.txt; : > $out
timeout 900 python3 -m pytest tests/test_gpu_lz4.py -x -q -m gpu 2>&1 | tail -8 | tee -a $out
for args in "--blocks 1024" "--blocks 3072" "--blocks 256" "--blocks 64 --lz4-path 2" "--block-size 1048576 --blocks 512" "--block-size 1048576 --blocks 128"; do
  for wv in 1 2; do
    echo "== $args --lz4-waves $wv" | tee -a $out
    timeout 300 python3 profiles/scripts/ab.py --steps 30 "--args=$args --lz4-waves $wv" prod 2>&1 | tail -1 | tee -a $out
  done
done
echo "== prof 1024" | tee -a $out
timeout 300 python3 profiles/scripts/ab.py --prof --steps 30 "--args=--blocks 1024 --lz4-waves 2" prod 2>&1 | tail -1 | tee -a $out
timeout 300 python3 profiles/scripts/ab.py --prof --steps 30 "--args=--blocks 1024 --lz4-waves 1" prod 2>&1 | tail -1 | tee -a $out
