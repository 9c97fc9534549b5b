"""ctypes binding of the C ABI in include/cryo_codec.h.

This module is the test/bench driver's view of the library: it adds nothing to
the codec, it only marshals numpy arrays and device pointers.  Names follow the
reference's domain (cryo blocks, methods, acceleration/level), see
reference compression.h:7-24.
"""
import ctypes as C

import numpy as np

from . import _loader

METHOD_LZ4 = 0   # COMP_LZ4, reference compression.h:9
METHOD_ZSTD = 1  # COMP_ZSTD, reference compression.h:10

OK = 0
E_ARG, E_HIP, E_NODEV, E_CORRUPT, E_DSTSIZE, E_UNSUPPORTED, E_NOMEM = -1, -2, -3, -4, -5, -6, -7
_ERR_NAMES = {0: "CRYO_OK", -1: "CRYO_E_ARG", -2: "CRYO_E_HIP", -3: "CRYO_E_NODEV",
              -4: "CRYO_E_CORRUPT", -5: "CRYO_E_DSTSIZE", -6: "CRYO_E_UNSUPPORTED",
              -7: "CRYO_E_NOMEM"}

# cryo_option (include/cryo_codec.h)
OPT_LZ4_DECODE_PATH, OPT_LZ4_INDEX_WALKERS, OPT_PIPE_MIN_BYTES, OPT_POOL_BYTES, OPT_ZSTD_DECODE_PATH = 1, 2, 3, 4, 5
OPT_LZ4_DECODE_WAVES = 9
OPT_WORKSPACE_KEEP_BYTES, OPT_WORKSPACE_MAX_BYTES, OPT_NUMA_LOCAL = 6, 7, 8
LZ4_PATH_AUTO, LZ4_PATH_RING, LZ4_PATH_INDEXED, LZ4_PATH_FEW_BLOCKS = 0, 1, 2, 3

DIST_WIDE, DIST_NARROW, DIST_INT4, DIST_RANDOM, DIST_ZEROS = range(5)
DIST_NAMES = ["wide", "narrow", "int4", "random", "zeros"]

# every symbol include/cryo_codec.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "cryo_codec_version", "cryo_codec_device_count", "cryo_codec_open", "cryo_codec_close",
    "cryo_codec_last_error", "cryo_codec_stream", "cryo_codec_sync", "cryo_codec_bound",
    "cryo_codec_set_option", "cryo_codec_get_option",
    "cryo_dev_alloc", "cryo_dev_free", "cryo_dev_upload", "cryo_dev_download", "cryo_dev_memset",
    "cryo_codec_compress_batch", "cryo_codec_decompress_batch", "cryo_codec_compress_block",
    "cryo_codec_decompress_block", "cryo_codec_compress_blocks", "cryo_codec_decompress_blocks",
    "cryo_codec_decompress_blocks_to",
    "cryo_multi_open", "cryo_multi_close", "cryo_multi_count", "cryo_multi_last_error",
    "cryo_multi_compress_blocks", "cryo_multi_decompress_blocks", "cryo_multi_decompress_blocks_to",
    "cryo_multi_decompress_blocks_keyed", "cryo_multi_set_option", "cryo_multi_pool_invalidate",
    "cryo_multi_get_transfer_counters",
    "cryo_codec_decompress_blocks_keyed", "cryo_codec_pool_invalidate", "cryo_codec_get_transfer_counters",
    "cryo_codec_trim", "cryo_multi_trim",
    "cryo_codec_synth_batch", "cryo_codec_checksum_batch",
    "cryo_codec_compare_batch", "cryo_checksum64", "cryo_codec_timer_start",
    "cryo_codec_timer_stop", "cryo_codec_get_counters",
]


class CryoError(RuntimeError):
    def __init__(self, code, what="", detail=""):
        self.code = code
        super().__init__("%s failed: %s (%d) %s" % (what, _ERR_NAMES.get(code, "?"), code, detail))


class TransferCounters(C.Structure):
    _fields_ = [("h2d_bytes", C.c_uint64), ("d2h_bytes", C.c_uint64), ("pool_hits", C.c_uint64),
                ("pool_misses", C.c_uint64), ("pool_blocks", C.c_uint64), ("pool_capacity", C.c_uint64)]


class Counters(C.Structure):
    _fields_ = [("blocks_compressed", C.c_uint64), ("blocks_decompressed", C.c_uint64),
                ("bytes_in", C.c_uint64), ("bytes_out", C.c_uint64), ("launches", C.c_uint64)]


_bound = False


def lib():
    """The loaded library with argtypes/restypes set."""
    global _bound
    L = _loader.load()
    if _bound:
        return L
    vp, u64, u32, i32, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_size_t
    L.cryo_codec_version.restype = C.c_char_p
    L.cryo_codec_device_count.restype = i32
    L.cryo_codec_open.argtypes = [i32, C.POINTER(vp)]
    L.cryo_codec_close.argtypes = [vp]
    L.cryo_codec_close.restype = None
    L.cryo_codec_last_error.argtypes = [vp]
    L.cryo_codec_last_error.restype = C.c_char_p
    L.cryo_codec_stream.argtypes = [vp]
    L.cryo_codec_stream.restype = vp
    L.cryo_codec_sync.argtypes = [vp]
    L.cryo_codec_set_option.argtypes = [vp, i32, C.c_int64]
    L.cryo_codec_get_option.argtypes = [vp, i32, C.POINTER(C.c_int64)]
    L.cryo_codec_bound.argtypes = [i32, sz]
    L.cryo_codec_bound.restype = sz
    L.cryo_dev_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.cryo_dev_free.argtypes = [vp, vp]
    L.cryo_dev_upload.argtypes = [vp, vp, vp, sz]
    L.cryo_dev_download.argtypes = [vp, vp, vp, sz]
    L.cryo_dev_memset.argtypes = [vp, vp, i32, sz]
    L.cryo_codec_compress_batch.argtypes = [vp, i32, i32, vp, u64, u32, u64, vp, u64, vp, vp]
    L.cryo_codec_decompress_batch.argtypes = [vp, i32, vp, vp, vp, vp, u64, u32, u64, vp]
    L.cryo_codec_compress_block.argtypes = [vp, i32, i32, vp, sz, vp, sz, C.POINTER(sz)]
    L.cryo_codec_decompress_block.argtypes = [vp, i32, vp, sz, vp, sz]
    L.cryo_codec_compress_blocks.argtypes = [vp, i32, i32, vp, sz, sz, vp, sz, vp]
    L.cryo_codec_decompress_blocks.argtypes = [vp, i32, vp, vp, sz, vp, sz, vp]
    L.cryo_codec_decompress_blocks_to.argtypes = [vp, i32, vp, vp, sz, vp, sz, vp]
    L.cryo_multi_open.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
    L.cryo_multi_close.argtypes = [vp]
    L.cryo_multi_close.restype = None
    L.cryo_multi_count.argtypes = [vp]
    L.cryo_multi_last_error.argtypes = [vp]
    L.cryo_multi_last_error.restype = C.c_char_p
    L.cryo_multi_compress_blocks.argtypes = [vp, i32, i32, vp, sz, sz, vp, sz, vp]
    L.cryo_multi_decompress_blocks.argtypes = [vp, i32, vp, vp, sz, vp, sz, vp]
    L.cryo_multi_decompress_blocks_to.argtypes = [vp, i32, vp, vp, sz, vp, sz, vp]
    L.cryo_multi_decompress_blocks_keyed.argtypes = [vp, i32, vp, vp, vp, sz, vp, sz, vp]
    L.cryo_multi_set_option.argtypes = [vp, i32, C.c_int64]
    L.cryo_multi_pool_invalidate.argtypes = [vp, u32, i32]
    L.cryo_codec_trim.argtypes = [vp]
    L.cryo_multi_trim.argtypes = [vp]
    L.cryo_multi_get_transfer_counters.argtypes = [vp, C.POINTER(TransferCounters)]
    L.cryo_codec_decompress_blocks_keyed.argtypes = [vp, i32, vp, vp, vp, sz, vp, sz, vp]
    L.cryo_codec_pool_invalidate.argtypes = [vp, u32, i32]
    L.cryo_codec_get_transfer_counters.argtypes = [vp, C.POINTER(TransferCounters)]
    L.cryo_codec_synth_batch.argtypes = [vp, u64, u64, u64, u64, u32, i32, vp, u64]
    L.cryo_codec_checksum_batch.argtypes = [vp, vp, u64, vp, u32, u64, vp]
    L.cryo_codec_compare_batch.argtypes = [vp, vp, u64, vp, u64, u32, u64, vp]
    L.cryo_checksum64.argtypes = [vp, sz]
    L.cryo_checksum64.restype = u64
    L.cryo_codec_timer_start.argtypes = [vp]
    L.cryo_codec_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    L.cryo_codec_get_counters.argtypes = [vp, C.POINTER(Counters)]
    _bound = True
    return L


def version():
    return lib().cryo_codec_version().decode()


def device_count():
    return lib().cryo_codec_device_count()


def bound(method, block_size):
    return lib().cryo_codec_bound(method, block_size)


def checksum64(data):
    a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
    return lib().cryo_checksum64(a.ctypes.data, a.nbytes)


class DeviceBuffer:
    """A hipMalloc'ed region owned by a Codec handle."""

    def __init__(self, codec, nbytes):
        self.codec = codec
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        codec._chk(codec.L.cryo_dev_alloc(codec.h, self.nbytes, C.byref(p)), "cryo_dev_alloc")
        self.ptr = p.value

    def free(self):
        if self.ptr:
            self.codec.L.cryo_dev_free(self.codec.h, self.ptr)
            self.ptr = None

    def upload(self, arr, offset=0):
        a = np.ascontiguousarray(arr)
        assert offset + a.nbytes <= self.nbytes
        self.codec._chk(self.codec.L.cryo_dev_upload(self.codec.h, self.ptr + offset, a.ctypes.data, a.nbytes),
                        "cryo_dev_upload")

    def download(self, nbytes=None, offset=0, dtype=np.uint8):
        nbytes = self.nbytes - offset if nbytes is None else int(nbytes)
        out = np.empty(nbytes, dtype=np.uint8)
        self.codec._chk(self.codec.L.cryo_dev_download(self.codec.h, out.ctypes.data, self.ptr + offset, nbytes),
                        "cryo_dev_download")
        return out.view(dtype)

    def memset(self, value=0):
        self.codec._chk(self.codec.L.cryo_dev_memset(self.codec.h, self.ptr, value, self.nbytes), "cryo_dev_memset")


class Codec:
    """One handle = one GPU + one HIP stream (include/cryo_codec.h)."""

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.cryo_codec_open(device, C.byref(h))
        if rc != OK:
            raise CryoError(rc, "cryo_codec_open(%d)" % device,
                            "no CPU fallback exists; a gfx950 GPU and the HIP runtime are required")
        self.h = h.value
        self.device = device

    def close(self):
        if self.h:
            self.L.cryo_codec_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, what):
        if rc != OK:
            raise CryoError(rc, what, (self.L.cryo_codec_last_error(self.h) or b"").decode())

    # -- plumbing --
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def sync(self):
        self._chk(self.L.cryo_codec_sync(self.h), "cryo_codec_sync")

    def set_option(self, option, value):
        self._chk(self.L.cryo_codec_set_option(self.h, option, int(value)), "cryo_codec_set_option")

    def get_option(self, option):
        v = C.c_int64()
        self._chk(self.L.cryo_codec_get_option(self.h, option, C.byref(v)), "cryo_codec_get_option")
        return v.value

    def timer_start(self):
        self._chk(self.L.cryo_codec_timer_start(self.h), "timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.L.cryo_codec_timer_stop(self.h, C.byref(ms)), "timer_stop")
        return ms.value

    def transfer_counters(self):
        t = TransferCounters()
        self._chk(self.L.cryo_codec_get_transfer_counters(self.h, C.byref(t)), "get_transfer_counters")
        return {f: getattr(t, f) for f, _ in TransferCounters._fields_}

    def trim(self):
        self._chk(self.L.cryo_codec_trim(self.h), "cryo_codec_trim")

    def pool_invalidate(self, key_hi=0, everything=False):
        self._chk(self.L.cryo_codec_pool_invalidate(self.h, key_hi, 1 if everything else 0), "pool_invalidate")

    def decompress_blocks_keyed(self, method, keys, comps, block_size):
        """host buffers in, host buffers out, through the device-resident pool; returns (list of arrays|None, statuses)"""
        n = len(comps)
        arrs = [np.ascontiguousarray(np.asarray(c, dtype=np.uint8)) for c in comps]
        src = (C.c_void_p * n)(*[a.ctypes.data if a.nbytes else None for a in arrs])
        szs = (C.c_uint32 * n)(*[a.nbytes for a in arrs])
        outs = [np.full(block_size, 0xA5, np.uint8) for _ in range(n)]
        dst = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        st = (C.c_int32 * n)()
        ks = (C.c_uint64 * n)(*[int(k) for k in keys])
        self._chk(self.L.cryo_codec_decompress_blocks_keyed(self.h, method, ks, src, szs, n, dst, block_size, st),
                  "decompress_blocks_keyed")
        st = np.array(list(st), np.int32)
        return [outs[i] if st[i] == 0 else None for i in range(n)], st

    def counters(self):
        c = Counters()
        self._chk(self.L.cryo_codec_get_counters(self.h, C.byref(c)), "get_counters")
        return {f: getattr(c, f) for f, _ in Counters._fields_}

    # -- device-resident batches (async on the handle's stream) --
    def synth_batch(self, seed, first_block, n, block_size, dist, d_dst, stride=None, block_step=1):
        stride = block_size if stride is None else stride
        self._chk(self.L.cryo_codec_synth_batch(self.h, seed, first_block, block_step, n, block_size, dist,
                                                d_dst.ptr, stride),
                  "synth_batch")

    def compress_batch(self, method, param, d_src, src_stride, block_size, n, d_dst, dst_stride, d_sizes, d_status):
        self._chk(self.L.cryo_codec_compress_batch(self.h, method, param, d_src.ptr, src_stride, block_size, n,
                                                   d_dst.ptr, dst_stride, d_sizes.ptr, d_status.ptr),
                  "compress_batch")

    def decompress_batch(self, method, d_src, d_off, d_sizes, d_dst, dst_stride, block_size, n, d_status):
        self._chk(self.L.cryo_codec_decompress_batch(self.h, method, d_src.ptr, d_off.ptr, d_sizes.ptr, d_dst.ptr,
                                                     dst_stride, block_size, n, d_status.ptr),
                  "decompress_batch")

    def checksum_batch(self, d_src, stride, n, d_sums, d_sizes=None, fixed_size=0):
        self._chk(self.L.cryo_codec_checksum_batch(self.h, d_src.ptr, stride, d_sizes.ptr if d_sizes else None,
                                                   fixed_size, n, d_sums.ptr), "checksum_batch")

    def compare_batch(self, d_a, a_stride, d_b, b_stride, block_size, n, d_mismatch):
        self._chk(self.L.cryo_codec_compare_batch(self.h, d_a.ptr, a_stride, d_b.ptr, b_stride, block_size, n,
                                                  d_mismatch.ptr), "compare_batch")

    # -- host convenience over batches: lists of numpy blocks in, lists out --
    def compress_blocks(self, method, param, blocks):
        """Compress equally sized host blocks as one device batch; returns list of uint8 arrays."""
        n = len(blocks)
        if n == 0:
            return []
        B = len(blocks[0])
        cap = bound(method, B)
        d_src, d_dst = self.alloc(n * B), self.alloc(n * cap)
        d_sizes, d_status = self.alloc(4 * n), self.alloc(4 * n)
        try:
            d_src.upload(np.concatenate([np.asarray(b, dtype=np.uint8) for b in blocks]))
            self.compress_batch(method, param, d_src, B, B, n, d_dst, cap, d_sizes, d_status)
            self.sync()
            st = d_status.download(dtype=np.int32)
            if (st != 0).any():
                raise CryoError(int(st[st != 0][0]), "compress_batch status")
            sizes = d_sizes.download(dtype=np.uint32)
            raw = d_dst.download()
            return [raw[i * cap:i * cap + int(sizes[i])].copy() for i in range(n)]
        finally:
            for b in (d_src, d_dst, d_sizes, d_status):
                b.free()

    def decompress_blocks(self, method, comps, block_size):
        """Decode host compressed blocks as one device batch; returns (list of arrays|None, statuses)."""
        n = len(comps)
        if n == 0:
            return [], np.zeros(0, np.int32)
        sizes = np.array([len(c) for c in comps], dtype=np.uint32)
        offs = np.zeros(n, dtype=np.uint64)
        pos = 0
        for i, c in enumerate(comps):
            offs[i] = pos
            pos += (len(c) + 15) & ~15
        packed = np.zeros(max(pos, 16), dtype=np.uint8)
        for i, c in enumerate(comps):
            packed[int(offs[i]):int(offs[i]) + len(c)] = np.asarray(c, dtype=np.uint8)
        d_src, d_off, d_sizes = self.alloc(packed.nbytes), self.alloc(8 * n), self.alloc(4 * n)
        d_dst, d_status = self.alloc(n * block_size), self.alloc(4 * n)
        try:
            d_src.upload(packed)
            d_off.upload(offs)
            d_sizes.upload(sizes)
            d_dst.memset(0xA5)
            self.decompress_batch(method, d_src, d_off, d_sizes, d_dst, block_size, block_size, n, d_status)
            self.sync()
            st = d_status.download(dtype=np.int32)
            raw = d_dst.download()
            outs = [raw[i * block_size:(i + 1) * block_size].copy() if st[i] == 0 else None for i in range(n)]
            return outs, st
        finally:
            for b in (d_src, d_off, d_sizes, d_dst, d_status):
                b.free()

    # -- single block through the host-buffer entry points (what the PG shim calls) --
    def compress_block(self, method, param, block):
        a = np.ascontiguousarray(np.asarray(block, dtype=np.uint8))
        cap = bound(method, a.nbytes)
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t()
        self._chk(self.L.cryo_codec_compress_block(self.h, method, param, a.ctypes.data, a.nbytes,
                                                   out.ctypes.data, cap, C.byref(n)), "compress_block")
        return out[:n.value].copy()

    def decompress_block(self, method, comp, block_size):
        a = np.ascontiguousarray(np.asarray(comp, dtype=np.uint8))
        out = np.empty(block_size, dtype=np.uint8)
        rc = self.L.cryo_codec_decompress_block(self.h, method, a.ctypes.data, a.nbytes, out.ctypes.data, block_size)
        if rc == E_CORRUPT:
            return None
        self._chk(rc, "decompress_block")
        return out
