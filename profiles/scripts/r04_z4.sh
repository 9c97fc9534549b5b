#!/bin/bash
# round 4: k_zchain4 with assembly loads + counted waits (prod) against the lane-per-block k_zchain (zlane)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_z4
{
timeout 1500 python3 -m pytest tests/test_gpu_zstd.py -x -q 2>&1 | tail -3
python3 profiles/scripts/ab.py --prof --steps 8 --args="--workload zstd_decode" zlane prod
python3 profiles/scripts/ab.py --steps 8 --args="--workload zstd_decode" zlane prod zlane prod
python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --blocks 8192 --block-size 1048576" zlane prod
python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --level 5 --blocks 16384" zlane prod
} 2>&1 | tee gpurun_out/r04_z4/out.txt
