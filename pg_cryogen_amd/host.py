"""ctypes binding of libcryo_host.so: the C host side that mirrors the reference's
compression.h / storage.c / cache.h surfaces plus the page-chain staging
(pg_cryogen_amd/host/*.h).  Driver for tests; the library itself is plain C."""
import ctypes as C
import os

from . import _loader

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libcryo_host.so")            # production: no test hooks exported
HOST_TEST_LIB_PATH = os.path.join(_HERE, "libcryo_host_test.so")  # + cryo_host_set_codec_ops (tests/conftest.py asks for it)

COMP_LZ4, COMP_ZSTD = 0, 1
(CRYO_ERR_SUCCESS, CRYO_ERR_DECOMPRESSION_FAILED, CRYO_ERR_WRONG_STARTING_BLOCK, CRYO_ERR_EMPTY_BLOCK,
 CRYO_ERR_CACHE_IS_FULL) = range(5)
InvalidBlockNumber = 0xFFFFFFFF
BLCKSZ = 8192

BOUND_FN = C.CFUNCTYPE(C.c_size_t, C.c_int, C.c_size_t)
COMPRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                          C.c_size_t, C.POINTER(C.c_uint32))
DECOMPRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_size_t,
                            C.c_void_p, C.c_size_t, C.POINTER(C.c_int32))
ERROR_HANDLER = C.CFUNCTYPE(None, C.c_int, C.c_char_p)


class CryoCodecOps(C.Structure):
    _fields_ = [("bound", BOUND_FN), ("compress_blocks", COMPRESS_FN), ("decompress_blocks", DECOMPRESS_FN),
                ("ctx", C.c_void_p), ("decompress_blocks_scatter", C.c_void_p),      # optional members: NULL in test doubles
                ("decompress_blocks_keyed", C.c_void_p), ("pool_invalidate", C.c_void_p)]


class CryoRel(C.Structure):
    _fields_ = [("relid", C.c_uint), ("handle", C.c_void_p), ("ops", C.c_void_p)]


class HeapTupleData(C.Structure):
    _fields_ = [("t_len", C.c_uint32), ("t_data", C.c_void_p)]


_libs = {}
_production = None   # None: by the environment (CRYO_HOST_TEST_HOOKS=1 selects the test build)


def use(production):
    """Select which build lib() returns from now on: the shipped libcryo_host.so (True), the build with the
    codec-double hook for CPU-only plumbing tests (False), or by the environment (None)."""
    global _production
    _production = production


def lib():
    prod = _production if _production is not None else os.environ.get("CRYO_HOST_TEST_HOOKS") != "1"
    if prod in _libs:
        return _libs[prod]
    _loader.load()  # libcryo_codec.so + one HIP runtime first
    path = HOST_LIB_PATH if prod else HOST_TEST_LIB_PATH
    if not os.path.exists(path):
        raise ImportError("pg_cryogen_amd: %s is missing; build with `make -C pg_cryogen_amd/host`" % path)
    L = C.CDLL(path)
    vp, u32, i32, sz = C.c_void_p, C.c_uint32, C.c_int, C.c_size_t
    L.cryo_compress.argtypes = [i32, vp, C.POINTER(sz)]
    L.cryo_compress.restype = vp
    L.cryo_decompress.argtypes = [i32, vp, sz, vp]
    L.cryo_decompress.restype = C.c_bool
    L.cryo_define_compression_gucs.restype = None
    if hasattr(L, "cryo_host_set_codec_ops"):   # test build only
        L.cryo_host_set_codec_ops.argtypes = [C.POINTER(CryoCodecOps)]
        L.cryo_host_set_codec_ops.restype = None
    L.cryo_host_codec_error.restype = C.c_char_p
    L.cryo_compat_set_error_handler.argtypes = [ERROR_HANDLER]
    L.cryo_compat_set_error_handler.restype = None
    L.cryo_init_page.argtypes = [vp]
    L.cryo_init_page.restype = None
    L.cryo_storage_insert.argtypes = [vp, C.POINTER(HeapTupleData)]
    L.cryo_storage_fetch.argtypes = [vp, i32, C.POINTER(HeapTupleData)]
    L.cryo_storage_fetch.restype = C.POINTER(HeapTupleData)
    L.cryo_storage_ntuples.argtypes = [vp]
    L.cryo_pages_needed.argtypes = [sz]
    L.cryo_stage_write_chain.argtypes = [C.POINTER(CryoRel), u32, i32, u32, vp, sz, C.POINTER(u32), i32,
                                         C.POINTER(i32)]
    L.cryo_stage_write_batch.argtypes = [C.POINTER(CryoRel), vp, i32, i32, u32, C.POINTER(u32)]
    L.cryo_stage_read_chain.argtypes = [C.POINTER(CryoRel), u32, C.POINTER(vp), C.POINTER(sz), C.POINTER(i32),
                                        C.POINTER(u32), C.POINTER(u32), u32, C.POINTER(u32)]
    L.cryo_memrel_create.restype = vp
    L.cryo_memrel_destroy.argtypes = [vp]
    L.cryo_memrel_destroy.restype = None
    L.cryo_memrel_bind.argtypes = [vp, C.c_uint, C.POINTER(CryoRel)]
    L.cryo_memrel_bind.restype = None
    L.cryo_memrel_reserve.argtypes = [vp]
    L.cryo_memrel_reserve.restype = u32
    L.cryo_memrel_set_frozen.argtypes = [vp, u32, C.c_bool]
    L.cryo_memrel_set_frozen.restype = None
    L.cryo_memrel_page.argtypes = [vp, u32]
    L.cryo_memrel_page.restype = vp
    L.cryo_memrel_nblocks.argtypes = [vp]
    L.cryo_memrel_nblocks.restype = u32
    L.cryo_init_cache.restype = None
    L.cryo_cache_configure.argtypes = [i32]
    L.cryo_cache_shutdown.restype = None
    L.cryo_read_data.argtypes = [C.POINTER(CryoRel), vp, u32, C.POINTER(i32)]
    L.cryo_read_data_batch.argtypes = [C.POINTER(CryoRel), C.POINTER(u32), i32, C.POINTER(i32), C.POINTER(i32)]
    L.cryo_scan_next_batch.argtypes = [C.POINTER(CryoRel), vp, i32, C.POINTER(u32), C.POINTER(i32), C.POINTER(i32)]
    L.cryo_seqscan_iter_create.restype = vp
    L.cryo_seqscan_iter_free.argtypes = [vp]
    L.cryo_seqscan_iter_free.restype = None
    L.cryo_seqscan_iter_next.argtypes = [vp]
    L.cryo_seqscan_iter_next.restype = u32
    L.cryo_seqscan_iter_exclude.argtypes = [vp, u32, C.c_bool]
    L.cryo_seqscan_iter_exclude.restype = C.c_bool
    L.cryo_seqscan_iter_reset.argtypes = [vp]
    L.cryo_seqscan_iter_reset.restype = None
    L.cryo_seqscan_iter_nranges.argtypes = [vp]
    L.cryo_cache_allocate.argtypes = [C.POINTER(CryoRel), u32]
    L.cryo_cache_release.argtypes = [i32]
    L.cryo_cache_release.restype = None
    L.cryo_cache_invalidate_relation.argtypes = [C.c_uint]
    L.cryo_cache_invalidate_relation.restype = None
    L.cryo_cache_get_pg_nblocks.argtypes = [i32]
    L.cryo_cache_get_pg_nblocks.restype = u32
    L.cryo_cache_get_data.argtypes = [i32]
    L.cryo_cache_get_data.restype = vp
    L.cryo_cache_get_xid.argtypes = [i32]
    L.cryo_cache_get_xid.restype = u32
    L.cryo_cache_err.argtypes = [i32]
    L.cryo_cache_err.restype = C.c_char_p
    L.cryo_host_transfer_counters.argtypes = [C.POINTER(C.c_uint64)] * 4
    L.cryo_host_transfer_counters.restype = None
    for n in ("cryo_cache_hits", "cryo_cache_misses", "cryo_cache_codec_calls"):
        getattr(L, n).restype = C.c_uint64
    _libs[prod] = L
    return L


def transfer_counters():
    """(h2d_bytes, d2h_bytes, pool_hits, pool_misses) of the bound GPU codec"""
    v = [C.c_uint64() for _ in range(4)]
    lib().cryo_host_transfer_counters(*[C.byref(x) for x in v])
    return tuple(x.value for x in v)


def set_int(name, value):
    C.c_int.in_dll(lib(), name).value = value


def get_int(name):
    return C.c_int.in_dll(lib(), name).value


def set_block_size(n):
    C.c_size_t.in_dll(lib(), "cryo_blcksz").value = n


def get_block_size():
    return C.c_size_t.in_dll(lib(), "cryo_blcksz").value
