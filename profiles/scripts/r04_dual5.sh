#!/bin/bash
# evidence for the two-wave indexed decoder: GPU suite, differential stress with the waves option rotated, then one wave
# against two per batch shape (bench.py, 30 steps, rocprofv3 kernel averages) and the barrier waits of block 0
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_lz4_dual.txt; : > $out
timeout 1200 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2 | tee -a $out
timeout 400 python3 tests/stress_gpu.py 200 ${1:-101} 2>&1 | tail -1 | tee -a $out
timeout 400 python3 tests/stress_gpu.py fuzz 150 ${2:-102} 2>&1 | tail -1 | tee -a $out
for args in "--blocks 64 --lz4-path 2" "--blocks 256" "--blocks 1024" "--blocks 2048" "--blocks 3072" "--blocks 4096" \
            "--block-size 1048576 --blocks 64" "--block-size 1048576 --blocks 512" "--block-size 1048576 --blocks 2048" \
            "--blocks 1024 --dist narrow" "--blocks 1024 --dist int4" "--blocks 1024 --dist zeros"; do
  for wv in 1 2; do
    printf "%-52s waves %d  " "$args" $wv | tee -a $out
    timeout 300 python3 profiles/scripts/ab.py --prof --steps 30 "--args=$args --lz4-waves $wv" prod 2>&1 | tail -1 | sed 's/^prod *//' | tee -a $out
  done
done
for args in "--blocks 1024" "--block-size 1048576 --blocks 512"; do
  echo "== barrier waits of block 0, $args (variant build -DCRYO_DUAL_PROF=1)" | tee -a $out
  CRYO_CODEC_LIB=profiles/variants_dualprof.so timeout 300 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 $args 2>&1 | grep "dual\]" | tail -2 | tee -a $out
done
