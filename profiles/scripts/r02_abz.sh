#!/bin/bash
# A/B of two builds: zstd level-1 encode rate (bench --workload zstd) + zstd encoder tests
cd "$GRAFT_REPO_ROOT"
run() { timeout 600 python3 bench.py --workload zstd --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['encode_GBps'], d['config']['decode_GBps'])"; }
cp pg_cryogen_amd/libcryo_codec.so /tmp/B.so
echo "B: $(run "$@")"
cp pg_cryogen_amd/libcryo_codec_A.so pg_cryogen_amd/libcryo_codec.so
echo "A: $(run "$@")"
cp /tmp/B.so pg_cryogen_amd/libcryo_codec.so
echo "B: $(run "$@")"
timeout 1500 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu -k "encode or corners" 2>&1 | tail -2
