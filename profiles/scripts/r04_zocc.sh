#!/bin/bash
# register-allocation hints for k_zexec (6 waves per SIMD) and k_zmove (8): variants against prod, zstd decode
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_zocc.txt; : > $out
for v in prod zocc6 zmocc8 zocc68 prod zocc6; do
  timeout 400 python3 profiles/scripts/ab.py --prof --steps 10 --args="--workload zstd_decode" $v 2>&1 | tail -1 | tee -a $out
done
