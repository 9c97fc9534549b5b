#!/bin/bash
# round 3, second GPU call: the suite with the pool tests, smoke, host API rates (incl. one block per call), zstd decode baseline
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_step2
O=gpurun_out/r03_step2
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 600 python3 profiles/host_api_rate.py 2>&1 | tee $O/host_api.txt
timeout 600 python3 bench.py --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-1800 | tee $O/bench_default.txt
timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1 | cut -c1-600 | tee $O/bench_zstd_decode.txt
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/zstats -o run -- python3 $R/bench.py --workload zstd_decode --no-cpu-baseline --steps 5 --warmup 1 > $R/$O/zstats.log 2>&1
cd $R
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03_step2/zstats/run_kernel_stats.csv')):
    print("%-44s calls %4s avg %10.1f us" % (r['Name'].split('(')[0][-44:], r['Calls'], float(r['AverageNs'])/1e3))
PY
