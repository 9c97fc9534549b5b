#!/bin/bash
# instruction mix of the decoder after the trims (SQ counters, three passes), then prod / before once more
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
timeout 900 bash profiles/scripts/pmc_sq.sh r04b_lz4_dec lz4_decode > gpurun_out/r04_valu2_sq.log 2>&1
out=gpurun_out/r04_valu2.txt; : > $out
for v in prod nodual; do
  timeout 400 python3 profiles/scripts/ab.py --prof --steps 10 --args="--workload zstd_decode" $v 2>&1 | tail -1 | tee -a $out
done
