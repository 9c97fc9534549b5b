"""ctypes access to the CPU oracle (oracle/libcryo_oracle.so) and, where present,
to the stock liblz4 / libzstd the reference links (reference Makefile:5).
Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libcryo_oracle.so")


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = self.L = C.CDLL(ORACLE_SO)
        vp, sz = C.c_void_p, C.c_size_t
        L.cryo_oracle_synth_block.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, vp]
        L.cryo_oracle_synth_block.restype = None
        L.cryo_oracle_lz4_bound.argtypes = [sz]
        L.cryo_oracle_lz4_bound.restype = sz
        L.cryo_oracle_lz4_compress.argtypes = [vp, sz, vp, sz, C.c_int]
        L.cryo_oracle_lz4_compress.restype = sz
        L.cryo_oracle_lz4_decompress.argtypes = [vp, sz, vp, sz]
        L.cryo_oracle_lz4_decompress.restype = C.c_long
        for name in ("cryo_oracle_zstd_decompress",):
            if hasattr(L, name):
                getattr(L, name).argtypes = [vp, sz, vp, sz]
                getattr(L, name).restype = C.c_long
        if hasattr(L, "cryo_oracle_zstd_bound"):
            L.cryo_oracle_zstd_bound.argtypes = [sz]
            L.cryo_oracle_zstd_bound.restype = sz
        if hasattr(L, "cryo_oracle_zstd_compress"):
            L.cryo_oracle_zstd_compress.argtypes = [vp, sz, vp, sz, C.c_int]
            L.cryo_oracle_zstd_compress.restype = sz

    def synth(self, seed, block, B, dist):
        a = np.empty(B, np.uint8)
        self.L.cryo_oracle_synth_block(seed, block, B, dist, a.ctypes.data)
        return a

    def lz4_bound(self, n):
        return self.L.cryo_oracle_lz4_bound(n)

    def lz4_compress(self, a, accel=1):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        cap = self.lz4_bound(a.nbytes)
        d = np.empty(max(cap, 1), np.uint8)
        r = self.L.cryo_oracle_lz4_compress(a.ctypes.data, a.nbytes, d.ctypes.data, cap, accel)
        return d[:r].copy()

    def lz4_decompress(self, c, cap, fill=0):
        c = np.ascontiguousarray(c, dtype=np.uint8)
        out = np.full(max(cap, 1), fill, np.uint8)
        r = self.L.cryo_oracle_lz4_decompress(c.ctypes.data, c.nbytes, out.ctypes.data, cap)
        return r, out[:cap]

    def zstd_decompress(self, c, cap, fill=0):
        c = np.ascontiguousarray(c, dtype=np.uint8)
        out = np.full(max(cap, 1), fill, np.uint8)
        r = self.L.cryo_oracle_zstd_decompress(c.ctypes.data, c.nbytes, out.ctypes.data, cap)
        return r, out[:cap]

    def zstd_compress(self, a, level=1):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        cap = self.L.cryo_oracle_zstd_bound(a.nbytes)
        d = np.empty(max(cap, 1), np.uint8)
        r = self.L.cryo_oracle_zstd_compress(a.ctypes.data, a.nbytes, d.ctypes.data, cap, level)
        return d[:r].copy()


class StockLibs:
    """liblz4.so.1 / libzstd.so.1 if they can be dlopen'ed (they are base-OS packages)."""

    def __init__(self):
        self.lz4 = self.zstd = None
        try:
            L = C.CDLL("liblz4.so.1")
            L.LZ4_versionString.restype = C.c_char_p
            L.LZ4_compress_fast.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
            L.LZ4_decompress_safe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
            L.LZ4_compressBound.argtypes = [C.c_int]
            self.lz4 = L
            self.lz4_version = L.LZ4_versionString().decode()
        except OSError:
            pass
        try:
            Z = C.CDLL("libzstd.so.1")
            Z.ZSTD_versionString.restype = C.c_char_p
            Z.ZSTD_compress.restype = C.c_size_t
            Z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
            Z.ZSTD_decompress.restype = C.c_size_t
            Z.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            Z.ZSTD_compressBound.restype = C.c_size_t
            Z.ZSTD_compressBound.argtypes = [C.c_size_t]
            Z.ZSTD_isError.argtypes = [C.c_size_t]
            self.zstd = Z
            self.zstd_version = Z.ZSTD_versionString().decode()
        except OSError:
            pass

    def lz4_compress(self, a, accel=1):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        cap = self.lz4.LZ4_compressBound(a.nbytes)
        d = np.empty(max(cap, 1), np.uint8)
        r = self.lz4.LZ4_compress_fast(a.ctypes.data, d.ctypes.data, a.nbytes, cap, accel)
        return d[:r].copy()

    def lz4_decompress(self, c, cap, fill=0):
        c = np.ascontiguousarray(c, dtype=np.uint8)
        out = np.full(max(cap, 1) + 64, fill, np.uint8)
        r = self.lz4.LZ4_decompress_safe(c.ctypes.data, out.ctypes.data, c.nbytes, cap)
        return r, out[:cap]

    def zstd_compress(self, a, level=1):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        cap = self.zstd.ZSTD_compressBound(a.nbytes)
        d = np.empty(max(cap, 1), np.uint8)
        r = self.zstd.ZSTD_compress(d.ctypes.data, cap, a.ctypes.data, a.nbytes, level)
        assert not self.zstd.ZSTD_isError(r)
        return d[:r].copy()

    def zstd_decompress(self, c, cap, fill=0):
        c = np.ascontiguousarray(c, dtype=np.uint8)
        out = np.full(max(cap, 1), fill, np.uint8)
        r = self.zstd.ZSTD_decompress(out.ctypes.data, cap, c.ctypes.data, c.nbytes)
        if self.zstd.ZSTD_isError(r):
            return -1, out[:cap]
        return int(r), out[:cap]
