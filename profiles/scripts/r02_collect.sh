#!/bin/bash
# round-2 evidence: bench line + rocprofv3 kernel stats + FETCH/WRITE counter passes per workload, SQ counters of the headline
bash profiles/collect.sh r02 lz4_decode > gpurun_out/collect_lz4_decode.log 2>&1
bash profiles/collect.sh r02 zstd_decode > gpurun_out/collect_zstd_decode.log 2>&1
bash profiles/collect.sh r02 lz4 > gpurun_out/collect_lz4.log 2>&1
bash profiles/collect.sh r02 zstd > gpurun_out/collect_zstd.log 2>&1
bash profiles/scripts/pmc_sq.sh r02_lz4_dec lz4_decode > gpurun_out/collect_sq.log 2>&1
for f in gpurun_out/collect_*.log; do tail -n 2 $f; done
