#!/bin/bash
# SQ instruction mix + stall counters of one bench workload, three counter passes (8 SQ slots per pass):
#   bash profiles/scripts/pmc_sq.sh <tag> <workload> [bench args]   -> gpurun_out/<tag>_sq.json
TAG=$1; WL=$2; shift; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${TAG}_sq; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/p1 -o run -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --steps 1 --warmup 0 "$@" > $OUT/log1.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -o run -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --steps 1 --warmup 0 "$@" > $OUT/log2.txt 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_ATOMIC --output-format csv -d $OUT/p3 -o run -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --steps 1 --warmup 0 "$@" > $OUT/log3.txt 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {}
for k, v in acc.items():
    if any(x in k for x in ("k_lz4", "k_z", "k_compare")):
        out[k] = {a: b / max(1, cnt[k][a]) for a, b in sorted(v.items())}
json.dump(out, open("$ROOT/gpurun_out/${TAG}_sq.json", "w"), indent=1)
for k, v in out.items():
    print(k, {a: "%.4g" % b for a, b in v.items()})
PY
