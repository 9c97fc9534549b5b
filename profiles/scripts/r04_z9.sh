#!/bin/bash
# round 4: chip-sized grids (grid-stride k_zhufw / k_zchain4 / fallback k_zhuf), after the drain fix
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_z9
ulimit -c 0
export HSA_ENABLE_COREDUMP=0
{
for b in 8192 16384; do timeout 120 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 1 --warmup 0 --blocks $b 2>&1 | grep -i "violation\|value" | cut -c1-120; done
timeout 900 python3 -m pytest tests/test_gpu_zstd.py tests/test_gpu_lz4.py -x -q 2>&1 | tail -3
timeout 300 python3 profiles/scripts/ab.py --prof --steps 8 --args="--workload zstd_decode" prod
timeout 300 python3 profiles/scripts/ab.py --steps 8 --args="--workload zstd_decode" zlane prod zlane prod
timeout 300 python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --blocks 8192 --block-size 1048576" zlane prod
timeout 300 python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --level 5 --blocks 16384" zlane prod
timeout 300 python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --blocks 4096" zlane prod
timeout 300 python3 profiles/scripts/ab.py --steps 6 --args="--workload zstd_decode --blocks 1024" zlane prod
timeout 300 python3 profiles/scripts/ab.py --steps 20 r03base prod
} 2>&1 | tee gpurun_out/r04_z9/out.txt
