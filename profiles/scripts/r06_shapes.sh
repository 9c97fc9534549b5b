#!/bin/bash
# round 6: mid-sized LZ4 / zstd decode batches with the cache flush SURVEY 8d prescribes (--flush), against
# the same shapes without it; where the byte-parallel few-blocks form stops paying (CRYO_LZ4_FEW_MAX, debug build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_shapes; mkdir -p $O; rm -f $O/*.txt
bash profiles/scripts/build_variant.sh dbg "-DCRYO_DEBUG" lz4_lat.hip > $O/build.log 2>&1
row() { # label, args...
  local label=$1; shift
  python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-58s %8.1f GB/s  %8.3f ms  frac %.4f' % ('$label', d['config']['blocks_per_gpu']*d['config']['block_size']/(r['avg_launch_ms']*1e-3)/1e9, r['avg_launch_ms'], r['frac']))"
}
for wl in lz4_decode zstd_decode; do
for f in on off; do
  for nb in 16384 4096 2048 1024 512 256 128 64 16 1; do row "$wl --blocks $nb --flush $f" --workload $wl --blocks $nb --flush $f; done
  for nb in 512 64 16 1; do row "$wl --block-size 1048576 --blocks $nb --flush $f" --workload $wl --block-size 1048576 --blocks $nb --flush $f; done
done; done > $O/shapes.txt
for m in 128 256 512; do
  for nb in 96 128 192 256 384 512; do
    if [ $nb -le $m ]; then CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_LZ4_FEW_MAX=$m row "few-blocks form up to $m: --blocks $nb --flush on" --workload lz4_decode --blocks $nb --flush on; fi
  done
done > $O/few_max.txt
cat $O/shapes.txt | head -60; cat $O/few_max.txt
