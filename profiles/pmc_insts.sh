#!/bin/bash
# instruction mix of one bench workload: bash profiles/pmc_insts.sh <workload> [bench args]
WL=${1:-lz4}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$WL; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $OUT -o run -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --steps 1 --warmup 0 "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k, {a: "%.3g" % b for a, b in sorted(v.items())})
PY
