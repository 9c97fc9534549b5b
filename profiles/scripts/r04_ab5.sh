#!/bin/bash
# round 4, fifth A/B: lane copies v2 alone (prod; match space as in round 3) against v1
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_ab5
O=gpurun_out/r04_ab5
{
timeout 900 python3 -m pytest tests/test_gpu_lz4.py -x -q 2>&1 | tail -2
python3 profiles/scripts/ab.py --prof copyv1 prod
python3 profiles/scripts/ab.py copyv1 prod copyv1 prod
for d in narrow int4 random; do python3 profiles/scripts/ab.py --args "--dist $d" copyv1 prod; done
python3 profiles/scripts/ab.py --args "--workload zstd_decode" --steps 6 copyv1 prod
} 2>&1 | tee $O/out.txt
