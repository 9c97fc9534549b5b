#!/bin/bash
# round-6 evidence on the final tree: the four workload collections (bench line, rocprofv3 kernel stats, FETCH / WRITE passes),
# SQ counters, the driver's invocation three times, batch shapes of both decoders, the encoders on the other distributions,
# host API, suite + smoke + soak
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
T=r06
timeout 900 bash profiles/collect.sh $T lz4_decode > gpurun_out/${T}_collect_lz4_decode.log 2>&1
timeout 900 bash profiles/collect.sh $T zstd_decode > gpurun_out/${T}_collect_zstd_decode.log 2>&1
timeout 900 bash profiles/collect.sh $T lz4 > gpurun_out/${T}_collect_lz4.log 2>&1
timeout 900 bash profiles/collect.sh $T zstd > gpurun_out/${T}_collect_zstd.log 2>&1
timeout 600 bash profiles/scripts/pmc_sq.sh ${T}_lz4_dec lz4_decode > gpurun_out/${T}_collect_sq.log 2>&1
for f in gpurun_out/${T}_collect_*.log; do tail -n 2 $f; done
for i in 1 2 3; do timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${T}_bench_driver_args_$i.json; done
python3 - <<'PY'
import json
for i in (1, 2, 3):
    d = json.load(open("gpurun_out/r06_bench_driver_args_%d.json" % i))
    c = d["cpu_baseline"]
    print("run %d: value %.1f GB/s, %.3f ms/step, frac %.4f | cpu 1 thread %.2f, %d pinned physical cores %.1f GB/s" %
          (i, d["value"], d["ms_per_step"], d["roofline"]["frac"], c["value"], c.get("cores_used", 0), c["all_cores_value"]))
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
cat $out
out=gpurun_out/${T}_zstd_decode_batch_shapes.txt; : > $out
z() { timeout 900 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 8 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); c = d['config']
    print('%-52s %9.1f GB/s  %9.3f ms  frac %.4f  ratio %s' % ('$*' or '(65536 x 128 KiB wide, level 1)', d['value'], d['ms_per_step'], d['roofline']['frac'], c['compression_ratio']))
except Exception as e:
    print('%-52s FAILED %s' % ('$*', e))" >> $out; }
z
z --blocks 16384
z --blocks 4096
z --blocks 1024
z --blocks 64
z --blocks 16
z --blocks 1
z --block-size 1048576 --blocks 8192
z --block-size 1048576 --blocks 512
z --block-size 1048576 --blocks 16
z --block-size 1048576 --blocks 1
z --blocks 16384 --level 3
z --blocks 16384 --level 5
z --blocks 16384 --level -5
z --dist narrow
z --dist int4
z --dist random
z --dist zeros
z --block-size 1048576 --blocks 8192 --dist narrow
cat $out
# the encoders on the shapes the reference's blocks have (narrow rows, zero gap), beside `wide`
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
timeout 400 python3 bench.py --workload mixed 2>/dev/null | tail -1 > gpurun_out/${T}_mixed_bench.json; cut -c1-300 gpurun_out/${T}_mixed_bench.json
# host API
{ for i in 1 2; do echo "== run $i"; HOST_API_REPS=7 timeout 900 python3 profiles/host_api_rate.py 2>&1 | grep -v "ONE block\|\[topology\] 0000\|amdgpu.ids"; done; } > gpurun_out/${T}_host_api.txt 2>&1; tail -8 gpurun_out/${T}_host_api.txt
# suite, smoke, soak, two ranks
out=gpurun_out/${T}_final_check.txt; : > $out
timeout 1700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee -a $out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $out
timeout 400 python3 tests/stress_gpu.py 150 61 2>&1 | tail -3 | tee -a $out
timeout 400 python3 tests/stress_gpu.py fuzz 100 62 2>&1 | tail -3 | tee -a $out
timeout 900 python3 bench.py --gpus 2 --steps 20 2>/dev/null | tail -1 | cut -c1-900 | tee -a $out
# LZ4 encoder's instruction mix, the crossover table of INTEGRATION.md with this round's encoders
timeout 600 bash profiles/scripts/pmc_sq.sh ${T}_lz4_enc lz4 > gpurun_out/${T}_collect_sq_enc.log 2>&1; tail -n 3 gpurun_out/${T}_collect_sq_enc.log | cut -c1-600
timeout 900 python3 profiles/crossover.py 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/${T}_crossover.txt; tail -40 gpurun_out/${T}_crossover.txt
