#!/bin/bash
# after the index changes (chain sweep, 255-runs, one walker below 16 KiB): LZ4 suite, the LZ4 decode collection and shapes again
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
T=r05
timeout 900 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -3
timeout 900 bash profiles/collect.sh $T lz4_decode > gpurun_out/${T}_collect_lz4_decode.log 2>&1; tail -2 gpurun_out/${T}_collect_lz4_decode.log
timeout 600 bash profiles/scripts/pmc_sq.sh ${T}_lz4_dec lz4_decode > gpurun_out/${T}_collect_sq.log 2>&1
for i in 1 2 3; do timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${T}_bench_driver_args_$i.json; done
python3 - <<'PY'
import json
for i in (1, 2, 3):
    d = json.load(open("gpurun_out/r05_bench_driver_args_%d.json" % i))
    print("run %d: value %.1f GB/s, %.3f ms/step, frac %.4f" % (i, d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
out=gpurun_out/${T}_lz4_decode_batch_shapes.txt; : > $out
b() { timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('%-52s %9.1f GB/s  %8.3f ms  frac %.4f  ratio %s' % ('$*' or '(headline: 65536 x 128 KiB wide)', d['value'], d['ms_per_step'], d['roofline']['frac'], c['compression_ratio']))" >> $out; }
b
b --blocks 131072
b --blocks 32768
b --blocks 16384
b --blocks 8192
b --blocks 4096
b --blocks 2048
b --blocks 1024
b --blocks 256
b --blocks 64
b --blocks 16
b --blocks 1
b --block-size 1048576 --blocks 8192
b --block-size 1048576 --blocks 2048
b --block-size 1048576 --blocks 512
b --block-size 1048576 --blocks 16
b --block-size 1048576 --blocks 1
b --dist narrow
b --dist int4
b --dist random
b --dist zeros
b --block-size 1048576 --blocks 8192 --dist narrow
b --block-size 1048576 --blocks 8192 --dist int4
b --block-size 1048576 --blocks 8192 --dist zeros
b --block-size 1048576 --blocks 512 --dist narrow
b --block-size 1048576 --blocks 64 --dist narrow
b --block-size 1048576 --blocks 16 --dist narrow
b --block-size 1048576 --blocks 1 --dist narrow
b --blocks 1024 --dist narrow
b --blocks 64 --dist int4
cat $out
out=gpurun_out/${T}_final_check.txt; : > $out
timeout 1700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee -a $out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $out
timeout 400 python3 tests/stress_gpu.py 150 81 2>&1 | tail -3 | tee -a $out
timeout 400 python3 tests/stress_gpu.py fuzz 100 82 2>&1 | tail -3 | tee -a $out
timeout 900 python3 bench.py --gpus 2 --steps 20 2>/dev/null | tail -1 | cut -c1-900 | tee -a $out
