#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run in the build container).

The reference (adjust/pg_cryogen) holds no codec arithmetic of its own: its
compression.c:61-123 calls liblz4 / libzstd, which it links un-vendored and
un-pinned (reference Makefile:5).  Its only test (sql/pg_cryogen.sql) pins SQL
round trips, never bytes.  The byte-level pins of this repo are therefore taken
from the real libraries installed in the build image:

    liblz4  1.9.3  (/usr/lib/x86_64-linux-gnu/liblz4.so.1)
    libzstd 1.4.8  (/usr/lib/x86_64-linux-gnu/libzstd.so.1), cross-checked == 1.4.9 (/opt/conda/lib)

called exactly as the reference calls them:
    LZ4_compress_fast(src, dst, B, LZ4_compressBound(B), accel)     compression.c:70-72
    LZ4_decompress_safe(src, dst, csize, B)                         compression.c:84
    ZSTD_compress(dst, ZSTD_compressBound(B), src, B, level)        compression.c:102-104
    ZSTD_decompress(dst, B, src, csize)                             compression.c:116

Inputs are the synthetic cryo blocks of include/cryo_synth.h (produced through the
oracle's generator; their bytes are themselves pinned here by raw_sha256).

Outputs (all small, committed):
    vectors.json     per (method, param, block size, distribution, block index):
                     raw_sha256, csize, comp_sha256
    streams.json     full compressed streams (base64) for the small cells, used to pin decoders
    adversarial.json malformed / edge streams with the library's accept-reject verdict
"""
import base64
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402

SIZES = [131072, 1 << 20]
LZ4_ACCELS = [0, 1, 2, 7, 50]
ZSTD_LEVELS = [-5, -1, 1, 2, 3, 4, 5, 7, 22]  # 4, 5, 7 added with the dfast / greedy kernels and the lazy oracle
DISTS = list(range(5))
BLOCKS = [0, 1, 2, 3]
SEED = 0


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def b64(a):
    return base64.b64encode(np.ascontiguousarray(a).tobytes()).decode()


def main():
    ora = oracle_lib.Oracle()
    stock = oracle_lib.StockLibs()
    assert stock.lz4 is not None and stock.zstd is not None, "needs liblz4.so.1 and libzstd.so.1"
    assert stock.lz4_version == "1.9.3", stock.lz4_version
    assert stock.zstd_version in ("1.4.8", "1.4.9"), stock.zstd_version
    # second zstd build for the cross-check
    z2 = None
    try:
        z2 = C.CDLL("/opt/conda/lib/libzstd.so.1")
        z2.ZSTD_versionString.restype = C.c_char_p
        z2.ZSTD_compress.restype = C.c_size_t
        z2.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    except OSError:
        pass

    vectors = {"lz4_version": stock.lz4_version, "zstd_version": stock.zstd_version,
               "zstd_crosscheck": z2.ZSTD_versionString().decode() if z2 else None,
               "seed": SEED, "cells": []}
    streams = {"note": "full compressed streams for small cells (base64)", "streams": []}
    for B in SIZES:
        for dist in DISTS:
            for blk in BLOCKS:
                raw = ora.synth(SEED, blk, B, dist)
                rsha = sha(raw)
                for accel in LZ4_ACCELS:
                    c = stock.lz4_compress(raw, accel)
                    vectors["cells"].append({"method": "lz4", "param": accel, "B": B, "dist": dist, "block": blk,
                                             "raw_sha256": rsha, "csize": int(len(c)), "comp_sha256": sha(c)})
                    if len(c) <= 20000 and blk < 2 and accel in (1, 50):
                        streams["streams"].append({"method": "lz4", "param": accel, "B": B, "dist": dist,
                                                   "block": blk, "raw_sha256": rsha, "data": b64(c)})
                for lvl in ZSTD_LEVELS:
                    if B == (1 << 20) and lvl == 22 and dist in (0, 3) and blk > 0:
                        continue  # level 22 on 1 MiB incompressible blocks: seconds each, keep one
                    c = stock.zstd_compress(raw, lvl)
                    if z2 is not None:
                        cap = len(raw) + (len(raw) >> 8) + 64
                        d = np.empty(cap, np.uint8)
                        r = z2.ZSTD_compress(d.ctypes.data, cap, raw.ctypes.data, len(raw), lvl)
                        assert np.array_equal(d[:r], c), "zstd 1.4.8 and 1.4.9 disagree"
                    vectors["cells"].append({"method": "zstd", "param": lvl, "B": B, "dist": dist, "block": blk,
                                             "raw_sha256": rsha, "csize": int(len(c)), "comp_sha256": sha(c)})
                    if len(c) <= 20000 and blk < 2 and lvl in (-5, 1, 3, 22):
                        streams["streams"].append({"method": "zstd", "param": lvl, "B": B, "dist": dist,
                                                   "block": blk, "raw_sha256": rsha, "data": b64(c)})
    # small-block streams that exercise the 16-bit LZ4 table and every zstd block type
    for B in (4096, 65546, 65547):
        for dist in (0, 1, 3):
            raw = ora.synth(SEED, 0, B, dist)
            c = stock.lz4_compress(raw, 1)
            vectors["cells"].append({"method": "lz4", "param": 1, "B": B, "dist": dist, "block": 0,
                                     "raw_sha256": sha(raw), "csize": int(len(c)), "comp_sha256": sha(c)})
            if len(c) <= 8192:
                streams["streams"].append({"method": "lz4", "param": 1, "B": B, "dist": dist, "block": 0,
                                           "raw_sha256": sha(raw), "data": b64(c)})
            for lvl in (1, 3, 22):
                z = stock.zstd_compress(raw, lvl)
                vectors["cells"].append({"method": "zstd", "param": lvl, "B": B, "dist": dist, "block": 0,
                                         "raw_sha256": sha(raw), "csize": int(len(z)), "comp_sha256": sha(z)})
                if len(z) <= 8192:
                    streams["streams"].append({"method": "zstd", "param": lvl, "B": B, "dist": dist, "block": 0,
                                               "raw_sha256": sha(raw), "data": b64(z)})

    # the deep zstd levels (round 3: lazy2 at 10, btlazy2 at 12 / 15, btopt at 13 / 16, btultra at 16 / 18, btultra2 at 19): sizes
    # and hashes only, after the older cells so that those keep their order
    for B, levels, blks in ((131072, (10, 12, 13, 16, 19), (0, 1)), (1 << 20, (12, 15, 16, 18, 19), (0,))):
        for dist in DISTS:
            for blk in blks:
                raw = ora.synth(SEED, blk, B, dist)
                for lvl in levels:
                    c = stock.zstd_compress(raw, lvl)
                    vectors["cells"].append({"method": "zstd", "param": lvl, "B": B, "dist": dist, "block": blk,
                                             "raw_sha256": sha(raw), "csize": int(len(c)), "comp_sha256": sha(c)})

    # ---- adversarial decode vectors: verdict of the real library on malformed input ----
    adv = {"note": "ok = library returned exactly B bytes; out_sha256 then pins the decoded bytes "
                   "(dst pre-filled with 0xA5)", "cases": []}
    rng = np.random.default_rng(20261002)
    B = 4096
    for method in ("lz4", "zstd"):
        for dist in (0, 1, 3):
            raw = ora.synth(SEED, 0, B, dist)
            base = stock.lz4_compress(raw, 1) if method == "lz4" else stock.zstd_compress(raw, 1)
            muts = [("valid", base)]
            muts.append(("truncated_1", base[:-1]))
            muts.append(("truncated_half", base[:len(base) // 2]))
            muts.append(("trailing_garbage", np.concatenate([base, np.array([1, 2, 3, 4, 5], np.uint8)])))
            muts.append(("empty", base[:0]))
            for k in range(40):
                m = base.copy()
                for _ in range(int(rng.integers(1, 4))):
                    m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
                muts.append(("flip_%d" % k, m))
            for k in range(10):
                m = base.copy()
                p = int(rng.integers(0, len(m) - 1))
                m[p] = 0
                m[p + 1] = 0
                muts.append(("zero2_%d" % k, m))
            if method == "zstd":
                m = base.copy(); m[0] ^= 0xFF
                muts.append(("bad_magic", m))
                m = base.copy(); m[4] ^= 0x04
                muts.append(("fhd_checksum_flag", m))
                m = base.copy(); m[5] = (int(m[5]) + 1) & 0xFF
                muts.append(("wrong_content_size", m))
            for name, m in muts:
                m = np.ascontiguousarray(m, dtype=np.uint8)
                if len(m) == 0:
                    ok, osha = False, None
                else:
                    if method == "lz4":
                        r, out = stock.lz4_decompress(m, B, fill=0xA5)
                    else:
                        r, out = stock.zstd_decompress(m, B, fill=0xA5)
                    ok = (r == B)
                    osha = sha(out) if ok else None
                adv["cases"].append({"method": method, "B": B, "dist": dist, "name": name, "ok": bool(ok),
                                     "out_sha256": osha, "data": b64(m)})

    for name, obj in (("vectors.json", vectors), ("streams.json", streams), ("adversarial.json", adv)):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(obj, f, indent=0, sort_keys=True)
            f.write("\n")
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")
    print("cells", len(vectors["cells"]), "streams", len(streams["streams"]), "adversarial", len(adv["cases"]))


if __name__ == "__main__":
    main()
