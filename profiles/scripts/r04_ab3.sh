#!/bin/bash
# round 4, third A/B: literal runs loaded from global memory into registers (prod) against reading them from the input ring (glit0)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_ab3
O=gpurun_out/r04_ab3
{
timeout 900 python3 -m pytest tests/test_gpu_lz4.py -x -q 2>&1 | tail -3
python3 profiles/scripts/ab.py --prof r03base glit0 prod
python3 profiles/scripts/ab.py r03base glit0 prod
python3 profiles/scripts/ab.py --args "--dist narrow" glit0 prod
python3 profiles/scripts/ab.py --args "--dist random" glit0 prod
python3 profiles/scripts/ab.py --args "--dist int4" glit0 prod
} 2>&1 | tee $O/out.txt
