#!/bin/bash
# round 6: is k_zhufw bound by its waves per CU?  (extra LDS per workgroup caps them: 7 -> 5 -> 4 -> 3); scattered-store micro-benchmark
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_zdec4; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/ub_scatter profiles/scripts/ub_scatter.hip 2>/dev/null && /tmp/ub_scatter | tee $O/ub_scatter.txt
for pad in 0 5120 10240 24576; do
  echo "== k_zhufw with $pad bytes of extra LDS per workgroup"
  CRYO_ZHUF_PAD=$pad CRYO_CODEC_LIB=$GRAFT_REPO_ROOT/profiles/variants_zdbg.so bash profiles/quick_stats.sh zstd_decode 2>&1 | grep -E "k_zhufw|k_zchain4|k_zexec " | cut -c1-120
  CRYO_ZHUF_PAD=$pad CRYO_CODEC_LIB=$GRAFT_REPO_ROOT/profiles/variants_zdbg.so python3 bench.py --workload zstd_decode --steps 10 --warmup 2 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   call %.1f GB/s %.3f ms' % (d['value'], d['ms_per_step']))"
done 2>&1 | tee $O/pad.txt
