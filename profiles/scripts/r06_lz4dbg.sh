#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06_lz4dbg; mkdir -p $O
bash profiles/scripts/build_variant.sh dbg "-DCRYO_DEBUG" lz4_enc2.hip > $O/build.txt 2>&1
for f in 0 1 2 3 4 7; do
  echo "== dbg=$f"
  CRYO_CODEC_LIB=profiles/variants_dbg.so CRYO_LZ4_ENC_DBG=$f timeout 300 python -m pytest tests/test_gpu_lz4.py -x -q -k "batch_kernel_corners" 2>&1 | grep -E "AssertionError: \(|passed|failed" | head -3
done
