#!/bin/bash
# round 4: zstd decode with the literal move in its own kernel (k_zmove) against round 3
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_z1
{
timeout 1200 python3 -m pytest tests/test_gpu_zstd.py -x -q 2>&1 | tail -2
python3 profiles/scripts/ab.py --prof --steps 8 --args="--workload zstd_decode" r03base prod
python3 profiles/scripts/ab.py --steps 8 --args="--workload zstd_decode" r03base prod r03base prod
python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --blocks 8192 --block-size 1048576" r03base prod
python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --level 5" r03base prod
} 2>&1 | tee gpurun_out/r04_z1/out.txt
