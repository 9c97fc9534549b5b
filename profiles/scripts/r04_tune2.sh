#!/bin/bash
# where one wave per block overtakes two: 1 536 ... 3 328 blocks of 128 KiB, 1 024 ... 3 328 of 1 MiB
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_tune2.txt; : > $out
for n in 1536 2048 2560 3072; do for wv in 1 2; do
  printf "%-58s " "--blocks $n --lz4-waves $wv" | tee -a $out
  timeout 300 python3 profiles/scripts/ab.py --prof --steps 20 "--args=--blocks $n --lz4-waves $wv" prod 2>&1 | tail -1 | sed 's/^prod *//' | tee -a $out
done; done
for n in 1024 2048 3072; do for wv in 1 2; do
  printf "%-58s " "--block-size 1048576 --blocks $n --lz4-waves $wv" | tee -a $out
  timeout 300 python3 profiles/scripts/ab.py --prof --steps 10 "--args=--block-size 1048576 --blocks $n --lz4-waves $wv" prod 2>&1 | tail -1 | sed 's/^prod *//' | tee -a $out
done; done
for d in narrow int4; do for wv in 1 2; do
  printf "%-58s " "--blocks 2048 --dist $d --lz4-waves $wv" | tee -a $out
  timeout 300 python3 profiles/scripts/ab.py --prof --steps 20 "--args=--blocks 2048 --dist $d --lz4-waves $wv" prod 2>&1 | tail -1 | sed 's/^prod *//' | tee -a $out
done; done
