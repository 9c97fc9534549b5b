#!/usr/bin/env python3
"""From how many blocks per call on does the device beat the library the reference links?  (VERDICT r04 item 7:
the unmodified reference hands over ONE block per call, pg_cryogen.c:726 and cache.c:178.)

For K = 1 .. 1024 blocks per call: wall time of the C host-buffer API (cryo_codec_{compress,decompress}_blocks, pageable
host memory both ways, PCIe included -- what host/compression.c's cryo_compress / cryo_decompress and the staging code
call) against stock liblz4 / libzstd on ONE host thread doing the same K blocks one after the other (the reference's
own path).  Prints the table and the smallest K from which the device call is the faster one.  Not the bench metric."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pg_cryogen_amd import Codec, METHOD_LZ4, METHOD_ZSTD, bound  # noqa: E402
import oracle_lib  # noqa: E402


def med(f, reps):
    v = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        v.append(time.perf_counter() - t0)
    return sorted(v)[len(v) // 2]


def main():
    stock = oracle_lib.StockLibs()
    ora = oracle_lib.Oracle()
    Ks = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024]
    with Codec(0) as c:
        L = c.L
        for B in (131072, 1048576):
            kmax = 1024 if B == 131072 else 256
            raws = [ora.synth(3, i, B, 0) for i in range(min(kmax, 64))]
            for method, param, name in ((METHOD_LZ4, 1, "lz4"), (METHOD_ZSTD, 1, "zstd")):
                enc1 = (lambda r: stock.lz4_compress(r, param)) if method == METHOD_LZ4 else (lambda r: stock.zstd_compress(r, param))
                dec1 = (lambda s: stock.lz4_decompress(s, B)) if method == METHOD_LZ4 else (lambda s: stock.zstd_decompress(s, B))
                comps = [enc1(r) for r in raws]
                cpu_enc = med(lambda: [enc1(r) for r in raws[:16]], 3) / 16      # seconds per block, one thread
                cpu_dec = med(lambda: [dec1(s) for s in comps[:16]], 3) / 16
                cap = bound(method, B)
                print("== %s, %d KiB blocks: stock library on one host thread %.3f ms (compress) / %.3f ms (decompress) per block" %
                      (name, B // 1024, cpu_enc * 1e3, cpu_dec * 1e3), flush=True)
                first_enc = first_dec = None
                for K in [k for k in Ks if k <= kmax]:
                    raw = np.concatenate([raws[i % len(raws)] for i in range(K)])
                    comp = np.zeros(K * cap, np.uint8)
                    sizes = np.zeros(K, np.uint32)
                    out = np.zeros(K * B, np.uint8)
                    st = np.zeros(K, np.int32)

                    def enc():
                        assert L.cryo_codec_compress_blocks(c.h, method, param, raw.ctypes.data, B, K, comp.ctypes.data, cap, sizes.ctypes.data) == 0
                    enc()
                    ptrs = (C.c_void_p * K)(*[comp.ctypes.data + i * cap for i in range(K)])

                    def dec():
                        assert L.cryo_codec_decompress_blocks(c.h, method, ptrs, sizes.ctypes.data, K, out.ctypes.data, B, st.ctypes.data) == 0
                    dec()
                    assert (st == 0).all() and np.array_equal(out, raw)
                    reps = 5 if K <= 64 else 3
                    ge, gd = med(enc, reps), med(dec, reps)
                    if first_enc is None and ge <= cpu_enc * K:
                        first_enc = K
                    if first_dec is None and gd <= cpu_dec * K:
                        first_dec = K
                    print("   K %5d: device compress %9.3f ms (host thread %9.3f)   device decompress %8.3f ms (host thread %8.3f)" %
                          (K, ge * 1e3, cpu_enc * K * 1e3, gd * 1e3, cpu_dec * K * 1e3), flush=True)
                print("   -> one device call beats ONE host thread from K = %s (compress), K = %s (decompress) blocks per call" %
                      (first_enc, first_dec), flush=True)


if __name__ == "__main__":
    main()
