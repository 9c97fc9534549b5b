#!/bin/bash
# round 4: k_zchain4 trimmed (mov_dpp without a zeroed destination, plain divergent branch for the second window read): 200 -> 174 instructions per two sequences
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_z12
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
{
timeout 900 python3 -m pytest tests/test_gpu_zstd.py -x -q 2>&1 | tail -2
timeout 300 python3 profiles/scripts/ab.py --prof --steps 8 --args="--workload zstd_decode" prod
timeout 300 python3 profiles/scripts/ab.py --steps 8 --args="--workload zstd_decode" zlane prod zlane prod
timeout 300 python3 profiles/scripts/ab.py --prof --steps 20 --args="--workload zstd_decode --blocks 1" prod
timeout 300 python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --level 5 --blocks 16384" prod
} 2>&1 | tee gpurun_out/r04_z12/out.txt
