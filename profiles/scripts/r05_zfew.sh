#!/bin/bash
# round 5: zstd decode of few frames per call: sequences executed byte-parallel (k_zlat_* + lat_copy.h) against k_zexec for
# every frame (CRYO_ZSTD_NO_FEW, debug build of zstd_pipe.hip)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_zstd.py tests/test_gpu_stress.py tests/test_gpu_host.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r05_zfew_tests.txt
out=gpurun_out/r05_zstd_few_frames.txt; : > $out
V=$PWD/profiles/variants_zpdbg.so
b() { # label, env..., then bench args after --
  local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env CRYO_CODEC_LIB=$V "${envs[@]}" timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 30 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-10s %-46s %9.2f GB/s  %8.3f ms' % ('$label', '$*', d['value'], d['ms_per_step']))" >> $out
}
for args in "--block-size 1048576 --blocks 1" "--block-size 1048576 --blocks 4" "--block-size 1048576 --blocks 16" "--block-size 1048576 --blocks 64" "--blocks 1" "--blocks 16" "--blocks 64" "--block-size 1048576 --blocks 1 --dist narrow" "--block-size 1048576 --blocks 16 --dist narrow" "--blocks 16 --level 5" "--block-size 1048576 --blocks 4 --level 19"; do
  b k_zexec CRYO_ZSTD_NO_FEW=1 -- $args
  b few A=1 -- $args
done
cat $out
timeout 300 python3 tests/stress_gpu.py 150 91 2>&1 | tail -2
timeout 300 python3 tests/stress_gpu.py fuzz 120 92 2>&1 | tail -2
