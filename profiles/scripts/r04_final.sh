#!/bin/bash
# round-4 evidence on the final kernels (two-wave decoder, trimmed copy engine, batches across index pieces):
# LZ4 decode + zstd decode collections, SQ counters, the driver's invocation three times, LZ4 batch shapes, suite + smoke + soak
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
T=r04
timeout 900 bash profiles/collect.sh $T lz4_decode > gpurun_out/${T}_collect_lz4_decode.log 2>&1
timeout 900 bash profiles/collect.sh $T zstd_decode > gpurun_out/${T}_collect_zstd_decode.log 2>&1
timeout 900 bash profiles/collect.sh $T lz4 > gpurun_out/${T}_collect_lz4.log 2>&1
timeout 600 bash profiles/scripts/pmc_sq.sh ${T}_lz4_dec lz4_decode > gpurun_out/${T}_collect_sq.log 2>&1
for f in gpurun_out/${T}_collect_*.log; do tail -n 2 $f; done
for i in 1 2 3; do timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${T}_bench_driver_args_$i.json; done
python3 - <<'PY'
import json
for i in (1, 2, 3):
    d = json.load(open("gpurun_out/r04_bench_driver_args_%d.json" % i))
    c = d["cpu_baseline"]
    print("run %d: value %.1f GB/s, %.3f ms/step, frac %.4f | cpu 1 thread %.2f, %d pinned physical cores %.1f GB/s" %
          (i, d["value"], d["ms_per_step"], d["roofline"]["frac"], c["value"], c.get("cores_used", 0), c["all_cores_value"]))
PY
out=gpurun_out/${T}_lz4_decode_batch_shapes.txt; : > $out
b() { timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('%-46s %9.1f GB/s  %8.3f ms  frac %.4f  ratio %s' % ('$*' or '(headline: 65536 x 128 KiB wide)', d['value'], d['ms_per_step'], d['roofline']['frac'], c['compression_ratio']))" >> $out; }
b
b --blocks 131072
b --blocks 32768
b --blocks 16384
b --blocks 4096
b --blocks 2048
b --blocks 1024
b --blocks 256
b --blocks 64
b --blocks 16
b --block-size 1048576 --blocks 8192
b --block-size 1048576 --blocks 2048
b --block-size 1048576 --blocks 512
b --block-size 1048576 --blocks 16
b --block-size 1048576 --blocks 1
b --dist narrow
b --dist int4
b --dist random
b --dist zeros
b --blocks 16384 --lz4-path 1
b --blocks 1024 --lz4-waves 1
b --block-size 1048576 --blocks 512 --lz4-waves 1
b --block-size 1048576 --blocks 8192 --lz4-path 1
b --block-size 1048576 --blocks 8192 --lz4-walkers 1
cat $out
out=gpurun_out/${T}_final_check.txt; : > $out
timeout 1700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee -a $out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $out
timeout 400 python3 tests/stress_gpu.py 150 51 2>&1 | tail -3 | tee -a $out
timeout 400 python3 tests/stress_gpu.py fuzz 100 52 2>&1 | tail -3 | tee -a $out
timeout 600 python3 bench.py --gpus 2 --steps 40 2>/dev/null | tail -1 | cut -c1-600 | tee -a $out
