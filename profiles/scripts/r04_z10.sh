#!/bin/bash
# timing experiment: what do the two (usually empty) fallback launches per tile cost the zstd decode call?
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_z10
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
{
timeout 300 python3 profiles/scripts/ab.py --steps 8 --args="--workload zstd_decode" zdebug zdebug
timeout 300 python3 profiles/scripts/ab.py --steps 8 --args="--workload zstd_decode" --env CRYO_ZSTD_SKIP_FALLBACKS=1 zdebug zdebug
} 2>&1 | tee gpurun_out/r04_z10/out.txt
