#!/bin/bash
# round 6, after the last kernel change (zstd fast finder's step positions): the encoders on every distribution again, the
# zstd compress + decompress line, suite + smoke + short soak on the final tree
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
T=r06
timeout 900 bash profiles/collect.sh $T zstd > gpurun_out/${T}_collect_zstd.log 2>&1; tail -n 2 gpurun_out/${T}_collect_zstd.log
out=gpurun_out/${T}_encode_distributions.txt; : > $out
for wl in lz4 zstd; do for dist in wide narrow int4 zeros random; do
  timeout 600 python3 bench.py --workload $wl --dist $dist --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); c = d['config']
    print('%-5s %-7s encode %8.2f GB/s  decode %8.1f GB/s  ratio %s' % ('$wl', '$dist', c['encode_GBps'], c['decode_GBps'], c['compression_ratio']))
except Exception as e:
    print('$wl $dist FAILED', e)" >> $out
done; done
for bs in 1048576; do for wl in lz4 zstd; do for dist in wide narrow; do
  timeout 600 python3 bench.py --workload $wl --dist $dist --block-size $bs --blocks 8192 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); c = d['config']
    print('%-5s %-7s 8192 x 1 MiB: encode %8.2f GB/s  decode %8.1f GB/s  ratio %s' % ('$wl', '$dist', c['encode_GBps'], c['decode_GBps'], c['compression_ratio']))
except Exception as e:
    print('$wl $dist 1MiB FAILED', e)" >> $out
done; done; done
cat $out
out=gpurun_out/${T}_final_check.txt; : > $out
timeout 1700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee -a $out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $out
timeout 400 python3 tests/stress_gpu.py 150 61 2>&1 | tail -3 | tee -a $out
timeout 400 python3 tests/stress_gpu.py fuzz 100 62 2>&1 | tail -3 | tee -a $out
timeout 900 python3 bench.py --gpus 2 --steps 20 2>/dev/null | tail -1 | cut -c1-900 | tee -a $out
