#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04_tests
{ timeout 1700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15; } | tee gpurun_out/r04_tests/out.txt
