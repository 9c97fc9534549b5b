#!/bin/bash
# round 3, first GPU call: the new index kernel (several walkers per block) -- suite, smoke, batch-shape sweep, kernel stats
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_step1
O=gpurun_out/r03_step1
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
b() { timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>&1 | tail -1 | python3 -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('$*', '| GB/s', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
except Exception as e:
    print('$*', '| FAILED', l[-300:])
"; }
{
b
b --lz4-walkers 2
b --blocks 32768
b --blocks 16384
b --blocks 16384 --lz4-path 1
b --blocks 4096
b --blocks 4096 --lz4-path 1
b --blocks 4096 --lz4-walkers 4
b --blocks 4096 --lz4-walkers 32
b --blocks 1024
b --blocks 1024 --lz4-path 1
b --blocks 64
b --blocks 64 --lz4-path 1
b --blocks 16
b --blocks 16 --lz4-path 1
b --block-size 1048576 --blocks 8192
b --block-size 1048576 --blocks 8192 --lz4-path 1
b --block-size 1048576 --blocks 8192 --lz4-walkers 1
b --block-size 1048576 --blocks 16
b --block-size 1048576 --blocks 16 --lz4-path 1
b --block-size 1048576 --blocks 1
b --block-size 1048576 --blocks 1 --lz4-path 1
b --dist narrow
b --dist int4
b --dist random
b --dist zeros
b --dist narrow --blocks 4096
b --dist int4 --blocks 4096
} 2>&1 | tee $O/sweep.txt
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o run -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 2 > $R/$O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats4k -o run -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 2 --blocks 4096 > $R/$O/stats4k.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats1m -o run -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 --block-size 1048576 --blocks 8192 > $R/$O/stats1m.log 2>&1
cd $R
for d in stats stats4k stats1m; do echo "== $d"; f=$(find $O/$d -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -d, -f1-8; done | tee $O/kernel_stats.txt
