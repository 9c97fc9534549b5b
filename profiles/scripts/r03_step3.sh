#!/bin/bash
# round 3, third GPU call: zstd encode size classes; small zstd batches through the pipeline vs the fused kernel
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r03_step3
O=gpurun_out/r03_step3
timeout 1500 python3 -m pytest tests/test_gpu_zstd.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt
cat > /tmp/zsmall.py <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle_lib
from pg_cryogen_amd import Codec, METHOD_ZSTD
ora = oracle_lib.Oracle(); stock = oracle_lib.StockLibs()
with Codec(0) as c:
    for B in (131072, 1 << 20):
        for n in (1, 2, 4, 8, 15, 16, 64):
            raws = [ora.synth(0, i, B, 0) for i in range(n)]
            comps = [stock.zstd_compress(r, 1) for r in raws]
            for _ in range(2):
                outs, st = c.decompress_blocks(METHOD_ZSTD, comps, B)
            assert (st == 0).all() and all(np.array_equal(o, r) for o, r in zip(outs, raws))
            # device-resident timing of the decode call alone
            sizes = np.array([len(x) for x in comps], np.uint32)
            offs = np.zeros(n, np.uint64); pos = 0
            for i, x in enumerate(comps):
                offs[i] = pos; pos += (len(x) + 15) & ~15
            packed = np.zeros(pos + 64, np.uint8)
            for i, x in enumerate(comps):
                packed[int(offs[i]):int(offs[i]) + len(x)] = x
            d_src, d_off, d_sz, d_dst, d_st = c.alloc(packed.nbytes), c.alloc(8 * n), c.alloc(4 * n), c.alloc(n * B), c.alloc(4 * n)
            d_src.upload(packed); d_off.upload(offs); d_sz.upload(sizes)
            ts = []
            for _ in range(6):
                c.timer_start(); c.decompress_batch(METHOD_ZSTD, d_src, d_off, d_sz, d_dst, B, B, n, d_st); ts.append(c.timer_stop())
            print("zstd decode %3d x %4d KiB  CRYO_ZSTD_PIPE=%s: %.3f ms" % (n, B // 1024, os.environ.get("CRYO_ZSTD_PIPE", "auto"), sorted(ts)[len(ts) // 2]))
            for x in (d_src, d_off, d_sz, d_dst, d_st): x.free()
PY
timeout 600 python3 /tmp/zsmall.py 2>&1 | tee $O/zsmall_auto.txt
CRYO_ZSTD_PIPE=1 timeout 600 python3 /tmp/zsmall.py 2>&1 | tee $O/zsmall_pipe.txt
CRYO_ZSTD_PIPE=0 timeout 600 python3 /tmp/zsmall.py 2>&1 | tee $O/zsmall_fused.txt
CRYO_ZSTD_STATS=1 timeout 300 python3 bench.py --workload zstd_decode --no-cpu-baseline --steps 1 --warmup 0 --blocks 14848 2>&1 | grep "zstd pipe" | head -3 | tee $O/zstats.txt
