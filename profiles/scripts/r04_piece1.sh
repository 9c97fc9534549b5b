#!/bin/bash
# batches that span index pieces: parity, then prod against the build before (variants_prev.so) per walkers setting
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
out=gpurun_out/r04_piece1.txt; : > $out
timeout 1200 python3 -m pytest tests/test_gpu_lz4.py tests/test_gpu_bench_workloads.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out
for args in "" "--blocks 1024 --lz4-walkers 16" "--blocks 1024 --lz4-walkers 32" "--blocks 1024 --lz4-walkers 64" \
            "--blocks 4096 --lz4-walkers 4" "--blocks 4096 --lz4-walkers 8" "--blocks 4096 --lz4-walkers 16" \
            "--blocks 16384 --lz4-walkers 1" "--blocks 16384 --lz4-walkers 2" "--blocks 16384 --lz4-walkers 4" \
            "--block-size 1048576 --blocks 512 --lz4-walkers 64" "--block-size 1048576 --blocks 8192 --lz4-walkers 4" "--block-size 1048576 --blocks 8192 --lz4-walkers 8"; do
  for v in prod prev; do
    printf "%-58s %-5s " "$args" $v | tee -a $out
    timeout 300 python3 profiles/scripts/ab.py --prof --steps 20 "--args=$args" $v 2>&1 | tail -1 | sed 's/^[a-z]* *//' | tee -a $out
  done
done
