"""Test double for the host side's CryoCodecOps (pg_cryogen_amd/host/compression.h): routes the
batch calls to the CPU oracle, and zstd compression to the stock libzstd the reference links.
It lets the PG-free plumbing tests (BASELINE config 1: "CPU-only liblz4 path") run without a GPU.
Test infrastructure only -- the product binds libcryo_codec.so."""
import ctypes as C

import numpy as np

import oracle_lib
from pg_cryogen_amd import host


class OracleCodecOps:
    def __init__(self):
        self.ora = oracle_lib.Oracle()
        self.stock = oracle_lib.StockLibs()
        self.compress_calls = 0
        self.decompress_calls = 0
        self.blocks_decompressed = 0
        self._bound = host.BOUND_FN(self.bound)
        self._comp = host.COMPRESS_FN(self.compress)
        self._decomp = host.DECOMPRESS_FN(self.decompress)
        self.ops = host.CryoCodecOps(self._bound, self._comp, self._decomp, None)

    def bound(self, method, n):
        return n + n // 255 + 16 if method == 0 else n + (n >> 8) + (((128 << 10) - n) >> 11 if n < (128 << 10) else 0)

    def compress(self, ctx, method, param, src, bs, n, dst, stride, out_size):
        self.compress_calls += 1
        for i in range(n):
            raw = np.ctypeslib.as_array(C.cast(src + i * bs, C.POINTER(C.c_uint8)), (bs,))
            if method == 0:
                c = self.ora.lz4_compress(raw, param)
            else:
                if self.stock.zstd is None:
                    return -6
                c = self.stock.zstd_compress(raw, param)
            C.memmove(dst + i * stride, c.ctypes.data, len(c))
            out_size[i] = len(c)
        return 0

    def decompress(self, ctx, method, srcs, sizes, n, dst, bs, status):
        self.decompress_calls += 1
        self.blocks_decompressed += n
        for i in range(n):
            comp = np.ctypeslib.as_array(C.cast(srcs[i], C.POINTER(C.c_uint8)), (sizes[i],))
            r, out = (self.ora.lz4_decompress if method == 0 else self.ora.zstd_decompress)(comp, bs)
            if r == bs:
                C.memmove(dst + i * bs, out.ctypes.data, bs)
                status[i] = 0
            else:
                status[i] = -4
        return 0
