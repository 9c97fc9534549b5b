#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_ab6
{
timeout 900 python3 -m pytest tests/test_gpu_zstd.py tests/test_gpu_lz4.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | cut -c1-330
timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 2>&1 | tail -1 | cut -c1-330
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_ab6/zstats -o run -- python3 $R/bench.py --workload zstd_decode --no-cpu-baseline --steps 5 --warmup 1 > $R/gpurun_out/r03_ab6/zstats.log 2>&1
cd $R
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03_ab6/zstats/run_kernel_stats.csv')):
    print("%-44s calls %4s avg %10.1f us" % (r['Name'].split('(')[0][-44:], r['Calls'], float(r['AverageNs'])/1e3))
PY
} 2>&1 | tee gpurun_out/r03_ab6/out.txt
