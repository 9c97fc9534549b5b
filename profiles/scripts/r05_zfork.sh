#!/bin/bash
# round 5: zstd decode, a tile's Huffman stage beside its sequence stage (CRYO_ZSTD_FORK_TILES = calls of at most that many
# tiles fork; 0 = never, round 4's order; debug build of zstd_pipe.hip so that the variable is read)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
out=gpurun_out/r05_zstd_fork.txt; : > $out
V=$PWD/profiles/variants_zpdbg.so
b() { # fork_tiles, bench args...
  local ft=$1; shift
  CRYO_CODEC_LIB=$V CRYO_ZSTD_FORK_TILES=$ft timeout 600 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('fork<=%-3s %-40s %9.1f GB/s  %8.3f ms' % ('$ft', '$*', d['value'], d['ms_per_step']))" >> $out
}
for args in "--blocks 1" "--blocks 16" "--blocks 64" "--blocks 256" "--blocks 1024" "--blocks 4096" "--blocks 12288" "--blocks 16384" "--blocks 24576" "--blocks 65536" "--block-size 1048576 --blocks 1" "--block-size 1048576 --blocks 16" "--block-size 1048576 --blocks 512" "--block-size 1048576 --blocks 8192" "--blocks 1024 --level 5" "--blocks 65536 --level 5"; do
  for ft in 0 2 99; do b $ft $args; done
done
cat $out
timeout 600 python3 -m pytest tests/test_gpu_zstd.py -x -q -m gpu 2>&1 | tail -3
