#!/bin/bash
# round 5, first look: store bandwidth ceiling; where the time goes on the zero-gap shapes; the single-pass decoder
# (k_lz4_dec_ring: the decoding wave parses the window it has staged) on the headline batch with its traffic and counters
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
./profiles/microbench/store_bw > gpurun_out/r05_store_bw.txt 2>&1; cat gpurun_out/r05_store_bw.txt
for d in zeros narrow int4; do
  echo "== $d"; bash profiles/quick_stats.sh lz4_decode --dist $d 2>&1 | tail -8
done > gpurun_out/r05_dist_kernels.txt 2>&1
cat gpurun_out/r05_dist_kernels.txt
bash profiles/collect.sh r05sp lz4_decode --lz4-path 1 > gpurun_out/r05sp_collect.log 2>&1; tail -3 gpurun_out/r05sp_collect.log
bash profiles/scripts/pmc_sq.sh r05sp_lz4_dec lz4_decode --lz4-path 1 > gpurun_out/r05sp_sq.log 2>&1; tail -4 gpurun_out/r05sp_sq.log
