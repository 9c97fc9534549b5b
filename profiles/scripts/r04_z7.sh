#!/bin/bash
# round 4: grids sized for the chip (grid-stride k_zhufw / k_zchain4 / fallback k_zhuf) against one workgroup per descriptor slot (zlane = r04_z4's prod would be ideal; here: before/after on one box via the committed numbers)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_z7
{
timeout 1500 python3 -m pytest tests/test_gpu_zstd.py -x -q 2>&1 | tail -3
python3 profiles/scripts/ab.py --prof --steps 8 --args="--workload zstd_decode" prod
python3 profiles/scripts/ab.py --steps 8 --args="--workload zstd_decode" prod prod
python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --blocks 8192 --block-size 1048576" prod
python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --level 5 --blocks 16384" prod
python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --blocks 4096" prod
python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --blocks 1024" prod
} 2>&1 | tee gpurun_out/r04_z7/out.txt
