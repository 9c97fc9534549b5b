#!/bin/bash
# round 6: zstd `fast` finder, short repeat-offset matches finished in ONE trip (everything they need rides with the table slots)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_zfl9; mkdir -p $O
V=${1:-zrep}
export CRYO_CODEC_LIB=$GRAFT_REPO_ROOT/profiles/variants_$V.so
timeout 1500 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_stress.py -x -q 2>&1 | tail -3
for a in "" "--dist narrow" "--dist int4" "--level -5" "--level 2" "--block-size 1048576 --blocks 8192"; do
  timeout 600 python3 bench.py --workload zstd --steps 2 --warmup 1 --no-cpu-baseline $a 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('%-44s encode %8.2f GB/s  decode %8.1f GB/s' % ('$a', c['encode_GBps'], c['decode_GBps']))"
done
