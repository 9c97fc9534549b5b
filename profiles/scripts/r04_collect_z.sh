#!/bin/bash
# round-4 evidence for the zstd workloads on the final kernels (k_zmove, k_zchain4), mixed, suite
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
T=r04
bash profiles/collect.sh $T zstd_decode > gpurun_out/${T}_collect_zstd_decode.log 2>&1
bash profiles/collect.sh $T zstd > gpurun_out/${T}_collect_zstd.log 2>&1
timeout 400 python3 bench.py --workload mixed 2>/dev/null | tail -1 > gpurun_out/${T}_mixed_bench.json
cut -c1-300 gpurun_out/${T}_mixed_bench.json
bash profiles/scripts/r04_zshapes.sh > gpurun_out/r04_zshapes.log 2>&1; tail -n 22 gpurun_out/r04_zshapes.log
out=gpurun_out/${T}_zstd_levels.txt; : > $out
for lvl in 1 3 5 7 10; do
  timeout 300 python3 bench.py --workload zstd --level $lvl --blocks 4096 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('level $lvl 128KiB x4096: encode %s GB/s decode %s GB/s ratio %s' % (c.get('encode_GBps'), c.get('decode_GBps'), c.get('compression_ratio')))" >> $out
done
cat $out
out=gpurun_out/${T}_final_check.txt; : > $out
timeout 1700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee -a $out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $out
timeout 400 python3 tests/stress_gpu.py 150 41 2>&1 | tail -3 | tee -a $out
timeout 400 python3 tests/stress_gpu.py fuzz 100 42 2>&1 | tail -3 | tee -a $out
timeout 600 python3 bench.py --gpus 2 --steps 40 2>/dev/null | tail -1 | cut -c1-600 | tee -a $out
