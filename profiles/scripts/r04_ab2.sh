#!/bin/bash
# round 4, second A/B: decoder with the far-match sources loaded first + counted wait; waves per workgroup 1 / 2 / 4 / 8
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_ab2
O=gpurun_out/r04_ab2
{
timeout 900 python3 -m pytest tests/test_gpu_lz4.py -x -q 2>&1 | tail -3
python3 profiles/scripts/ab.py --prof r03base prod wpb1 wpb2 wpb8
python3 profiles/scripts/ab.py r03base prod wpb1 wpb2 wpb8
} 2>&1 | tee $O/out.txt
