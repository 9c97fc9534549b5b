#!/bin/bash
# A/B of two builds on the same box: pg_cryogen_amd/libcryo_codec.so (B) against libcryo_codec_A.so (A)
cd "$GRAFT_REPO_ROOT"
run() { timeout 600 python3 bench.py --steps 80 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['ms_per_step'])"; }
cp pg_cryogen_amd/libcryo_codec.so /tmp/B.so
echo "B: $(run "$@")"
cp pg_cryogen_amd/libcryo_codec_A.so pg_cryogen_amd/libcryo_codec.so
echo "A: $(run "$@")"
cp /tmp/B.so pg_cryogen_amd/libcryo_codec.so
echo "B: $(run "$@")"
cp pg_cryogen_amd/libcryo_codec_A.so pg_cryogen_amd/libcryo_codec.so
echo "A: $(run "$@")"
cp /tmp/B.so pg_cryogen_amd/libcryo_codec.so
CRYO_LZ4_INDEX_MIN=1 timeout 900 python -m pytest tests/test_gpu_lz4.py -x -q -m gpu 2>&1 | tail -2
