#!/bin/bash
# after the few-blocks path: host API rates, headline unchanged?, suite + soak
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 600 python3 profiles/host_api_rate.py > gpurun_out/r03_host_api.txt 2>&1; cat gpurun_out/r03_host_api.txt
for i in 1 2; do timeout 300 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
bash profiles/scripts/r03_check.sh 120 120
