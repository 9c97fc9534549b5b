#!/bin/bash
# round 6: k_zexec reads the Huffman walkers' pieces in place (no k_zmove) -- correctness, then A/B against the move
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_zdec1; mkdir -p $O; rm -f $O/ab.txt
timeout 1700 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_stress.py tests/test_gpu_bench_workloads.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
bash profiles/scripts/build_variant.sh dbg "-DCRYO_DEBUG" zstd_pipe.hip > $O/build.txt 2>&1
row() { local label=$1; shift; env "$@" | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-44s %8.1f GB/s  %8.3f ms' % ('$label', d['value'], r['avg_launch_ms']))"; }
for rep in 1 2; do
row "in place (production)" python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null
row "k_zmove for every block" CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZSTD_MOVE=1 python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null
done
row "in place, 8192 x 1 MiB" python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline --block-size 1048576 --blocks 8192 2>/dev/null
row "move, 8192 x 1 MiB" CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZSTD_MOVE=1 python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline --block-size 1048576 --blocks 8192 2>/dev/null
row "in place, level 5 16384" python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline --level 5 --blocks 16384 2>/dev/null
row "move, level 5 16384" CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZSTD_MOVE=1 python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline --level 5 --blocks 16384 2>/dev/null
row "in place, narrow" python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline --dist narrow 2>/dev/null
row "move, narrow" CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_ZSTD_MOVE=1 python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline --dist narrow 2>/dev/null
bash profiles/quick_stats.sh zstd_decode --steps 5 2>&1 | grep "k_z" | head -12
