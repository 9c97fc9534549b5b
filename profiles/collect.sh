#!/bin/bash
# Evidence recipe for one round (run on the GPU box through gpurun from the repo root):
#   bash profiles/collect.sh r02 [lz4_decode|zstd_decode|lz4|zstd]
# Writes gpurun_out/<tag>_<workload>/{bench.json, stats/, fetch/, write/}; profiles/summarize.py turns
# those into the files committed under profiles/.  Counter passes are separate runs without tracing
# (MI355X_MICROARCH.md, HBM section).
TAG=${1:-r02}; WL=${2:-lz4_decode}; shift; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${TAG}_${WL}; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --workload $WL --steps 60 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --steps 20 --warmup 2 "$@" > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/write.log 2>&1
cd $ROOT
tail -1 $OUT/bench.json | cut -c1-600; ls $OUT
