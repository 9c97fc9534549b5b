#!/bin/bash
# quick per-kernel timing of one bench workload: bash profiles/quick_stats.sh <workload> [extra bench args]
WL=${1:-zstd_decode}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/quick_$WL; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --steps 5 --warmup 1 "$@" > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-400
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls %4s avg %10.3f ms %6s%%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
