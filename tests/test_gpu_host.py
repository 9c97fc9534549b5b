"""GPU plumbing tests: the SHIPPED host library (libcryo_host.so: compression.h mirror, staging, cache) driving the
real HIP codec through the C ABI -- no test double, no test hook.  Mirrors tests/test_host_plumbing.py."""
import ctypes as C
import struct

import numpy as np
import pytest

import oracle_lib
from pg_cryogen_amd import host
from test_host_plumbing import _load, fetch_rows

pytestmark = pytest.mark.gpu


@pytest.fixture()
def HG():
    host.use(production=True)                  # libcryo_host.so: binds libcryo_codec.so on GPU 0, exports no hook
    L = host.lib()
    assert not hasattr(L, "cryo_host_set_codec_ops")
    errors = []
    handler = host.ERROR_HANDLER(lambda lvl, msg: errors.append((lvl, msg.decode())) if lvl >= 20 else None)
    L.cryo_compat_set_error_handler(handler)
    host.set_block_size(131072)
    L.cryo_define_compression_gucs()
    L.cryo_cache_configure(16)
    yield L, errors
    L.cryo_cache_shutdown()
    L.cryo_compat_set_error_handler(host.ERROR_HANDLER(0))
    host.set_block_size(1 << 20)
    host.use(production=None)


def test_gpu_copy_10k_int4_lz4_roundtrip(HG, oracle):
    L, errors = HG
    rows = [struct.pack("<i", i) for i in range(1, 10001)]
    mem, rel, blocks, firsts = _load(L, rows, 1, host.COMP_LZ4, batch=16)
    assert len(blocks) == 35 and not errors
    # the bytes in the pages are exactly what liblz4 1.9.3 would have produced (oracle-pinned)
    for i in (0, 17, 34):
        comp = C.c_void_p()
        csize = C.c_size_t()
        method = C.c_int()
        xid = C.c_uint32()
        chain = (C.c_uint32 * 64)()
        n = C.c_uint32()
        assert L.cryo_stage_read_chain(C.byref(rel), firsts[i], C.byref(comp), C.byref(csize), C.byref(method),
                                       C.byref(xid), chain, 64, C.byref(n)) == 0
        got = np.ctypeslib.as_array(C.cast(comp, C.POINTER(C.c_uint8)), (csize.value,)).copy()
        exp = oracle.lz4_compress(np.frombuffer(blocks[i], np.uint8), 1)
        assert np.array_equal(got, exp)
    k = len(firsts)
    res, errs = (C.c_int * k)(), (C.c_int * k)()
    L.cryo_cache_configure(40)
    assert L.cryo_read_data_batch(C.byref(rel), (C.c_uint32 * k)(*firsts), k, res, errs) == 0
    assert L.cryo_cache_codec_calls() == 1           # read-ahead of 35 chains = one GPU batch
    ids = []
    for e in res:
        ids += [struct.unpack("<i", r)[0] for r in fetch_rows(L, L.cryo_cache_get_data(e))]
    assert len(ids) == 10000 and sum(ids) / len(ids) == 5000.5 and ids[:10] == list(range(1, 11))
    L.cryo_memrel_destroy(mem)


def test_gpu_unchanged_am_scan_decodes_k_blocks_per_codec_call(HG, oracle):
    """the reference's own scan loop (pg_cryogen.c:250-275) -- one iterator pop, one cryo_read_data
    per block -- over the 35-block table of config 1.  A miss loads the next block starts with the same device call
    (pg_cryogen.gpu_readahead_blocks): codec calls = ceil(35 / K), bytes of every block == what was written."""
    L, errors = HG
    rows = [struct.pack("<i", i) for i in range(1, 10001)]
    mem, rel, blocks, firsts = _load(L, rows, 1, host.COMP_LZ4, batch=16, relid=4343)
    assert len(blocks) == 35
    guc = C.c_int.in_dll(L, "cryo_gpu_readahead_blocks_guc")
    for k in (8, 1, 5):
        L.cryo_cache_configure(16)                       # the reference's cache size (cache.c:17)
        guc.value = k
        it = L.cryo_seqscan_iter_create()
        calls0 = L.cryo_cache_codec_calls()
        got = []
        while True:
            b = L.cryo_seqscan_iter_next(it)
            if L.cryo_memrel_nblocks(mem) <= b:
                break
            e = C.c_int(-1)
            err = L.cryo_read_data(C.byref(rel), it, b, C.byref(e))
            if err == host.CRYO_ERR_EMPTY_BLOCK:
                continue
            assert err == host.CRYO_ERR_SUCCESS, (b, err)
            got.append(bytes(np.ctypeslib.as_array(C.cast(L.cryo_cache_get_data(e.value), C.POINTER(C.c_uint8)), (host.get_block_size(),))))
        L.cryo_seqscan_iter_free(it)
        assert len(got) == 35 and [i for i in range(35) if got[i] != blocks[i]] == [], k
        assert L.cryo_cache_codec_calls() - calls0 == -(-35 // k), k
    guc.value = 8
    assert not errors
    L.cryo_memrel_destroy(mem)


def test_gpu_reads_chains_written_by_stock_libraries(HG, oracle):
    """a stock pg_cryogen wrote these pages with liblz4 / libzstd; the GPU decoder reads them"""
    L, errors = HG
    stock = oracle_lib.StockLibs()
    if stock.zstd is None or stock.lz4 is None:
        pytest.skip("stock libraries not loadable")
    B = 131072
    mem = L.cryo_memrel_create()
    rel = host.CryoRel()
    L.cryo_memrel_bind(mem, 99, C.byref(rel))
    raws, firsts = [], []
    for dist in range(5):
        for method, comp in ((1, lambda r: stock.zstd_compress(r, 1)), (1, lambda r: stock.zstd_compress(r, 22)),
                             (0, lambda r: stock.lz4_compress(r, 1))):
            if method == 1 and dist in (0, 3) and len(raws) % 3 == 1:
                comp = lambda r: stock.zstd_compress(r, 3)   # level 22 on incompressible data is slow
            raw = oracle.synth(5, dist, B, dist)
            c = comp(raw)
            fb = L.cryo_memrel_reserve(mem)
            chain = (C.c_uint32 * 64)()
            np_ = C.c_int()
            assert L.cryo_stage_write_chain(C.byref(rel), fb, method, 1234, c.ctypes.data, len(c), chain, 64,
                                            C.byref(np_)) == 0
            raws.append(raw)
            firsts.append(fb)
    k = len(firsts)
    res, errs = (C.c_int * k)(), (C.c_int * k)()
    assert L.cryo_read_data_batch(C.byref(rel), (C.c_uint32 * k)(*firsts), k, res, errs) == 0
    assert L.cryo_cache_codec_calls() == 2           # one batch per method
    for raw, e in zip(raws, res):
        got = np.ctypeslib.as_array(C.cast(L.cryo_cache_get_data(e), C.POINTER(C.c_uint8)), (B,))
        assert np.array_equal(got, raw)
    assert not errors
    L.cryo_memrel_destroy(mem)


def test_gpu_compression_h_surface(HG, oracle):
    L, errors = HG
    raw = oracle.synth(0, 3, 131072, 0)
    n = C.c_size_t(0)
    p = L.cryo_compress(host.COMP_LZ4, raw.ctypes.data, C.byref(n))
    comp = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,)).copy()
    assert np.array_equal(comp, oracle.lz4_compress(raw, 1))
    out = np.zeros(131072, np.uint8)
    assert L.cryo_decompress(host.COMP_LZ4, comp.ctypes.data, len(comp), out.ctypes.data) is True
    assert np.array_equal(out, raw)
    assert L.cryo_decompress(host.COMP_LZ4, comp.ctypes.data, len(comp) - 9, out.ctypes.data) is False
    assert not errors
    # zstd at the reference's default level 1 (compression.c:18): bytes identical to libzstd 1.4.8
    p = L.cryo_compress(host.COMP_ZSTD, raw.ctypes.data, C.byref(n))
    zc = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,)).copy()
    assert np.array_equal(zc, oracle.zstd_compress(raw, 1))
    assert L.cryo_decompress(host.COMP_ZSTD, zc.ctypes.data, len(zc), out.ctypes.data) is True
    assert np.array_equal(out, raw) and not errors
    # a `lazy2` level (9) has a kernel: same bytes as the library
    host.set_int("zstd_compression_level_guc", 9)
    p = L.cryo_compress(host.COMP_ZSTD, raw.ctypes.data, C.byref(n))
    z9 = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,)).copy()
    assert np.array_equal(z9, oracle.zstd_compress(raw, 9)) and not errors
    # the optimal-parser levels (btopt .. btultra2, 13..22 at this size) have kernels too: the GUC's whole range compresses
    # on the GPU, with the library's own bytes
    stock = oracle_lib.StockLibs()
    for lvl in (13, 22):
        host.set_int("zstd_compression_level_guc", lvl)
        p = L.cryo_compress(host.COMP_ZSTD, raw.ctypes.data, C.byref(n))
        z = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,)).copy()
        assert np.array_equal(z, oracle.zstd_compress(raw, lvl)) and not errors
        if stock.zstd is not None:
            assert np.array_equal(z, stock.zstd_compress(raw, lvl))
        assert L.cryo_decompress(host.COMP_ZSTD, z.ctypes.data, len(z), out.ctypes.data) is True and np.array_equal(out, raw)
    host.set_int("zstd_compression_level_guc", 1)


def test_c_host_batch_api_strides_and_statuses(codec, oracle):
    """cryo_codec_{compress,decompress}_blocks straight through the C ABI: exact and padded output strides
    (bulk / per-block copy paths), mixed valid / corrupt inputs, reuse of the handle's grow-only buffers"""
    import ctypes as C
    from pg_cryogen_amd import METHOD_LZ4, METHOD_ZSTD, bound
    L = codec.L
    for method, name in ((METHOD_LZ4, "lz4"), (METHOD_ZSTD, "zstd")):
        for B, n in ((131072, 5), (20000, 33), (131072, 2)):
            raw = np.concatenate([oracle.synth(9, i, B, i % 5) for i in range(n)])
            cap = bound(method, B)
            for stride in (cap, cap + 4097, cap + 64):
                comp = np.zeros(n * stride, np.uint8)
                sizes = np.zeros(n, np.uint32)
                rc = L.cryo_codec_compress_blocks(codec.h, method, 1, raw.ctypes.data, B, n, comp.ctypes.data, stride,
                                                  sizes.ctypes.data)
                assert rc == 0, (name, B, stride, rc)
                for i in range(n):
                    exp = oracle.lz4_compress(raw[i * B:(i + 1) * B], 1) if method == METHOD_LZ4 else \
                        oracle.zstd_compress(raw[i * B:(i + 1) * B], 1)
                    assert int(sizes[i]) == len(exp) and np.array_equal(comp[i * stride:i * stride + len(exp)], exp), (name, B, stride, i)
            bufs = [comp[i * stride:i * stride + int(sizes[i])].copy() for i in range(n)]
            bufs[n // 2] = bufs[n // 2][:len(bufs[n // 2]) // 2].copy()        # truncated -> corrupt
            ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
            szs = np.array([len(b) for b in bufs], np.uint32)
            out = np.zeros(n * B, np.uint8)
            st = np.zeros(n, np.int32)
            rc = L.cryo_codec_decompress_blocks(codec.h, method, ptrs, szs.ctypes.data, n, out.ctypes.data, B, st.ctypes.data)
            assert rc == 0
            for i in range(n):
                if i == n // 2:
                    assert st[i] != 0
                else:
                    assert st[i] == 0 and np.array_equal(out[i * B:(i + 1) * B], raw[i * B:(i + 1) * B]), (name, B, i)


def test_gpu_seqscan_read_ahead_through_hip_codec(HG):
    """SURVEY.md 8f-3 on the device: cryo_scan_next_batch pops block starts in the reference's order
    (scan_iterator.c:55-127), never offers continuation pages, and decodes K chains -- multi-page and
    single-page, lz4 and zstd mixed in one table -- with one HIP codec call per method and batch."""
    from test_host_plumbing import pack_rows
    L, errors = HG
    L.cryo_cache_configure(32)
    rng = np.random.default_rng(5)
    mem = L.cryo_memrel_create()
    rel = host.CryoRel()
    L.cryo_memrel_bind(mem, 7, C.byref(rel))
    all_rows, firsts, methods = [], [], []
    for b in range(11):
        if b % 2:   # incompressible rows -> ~15-page chain
            rows = [struct.pack("<i", 1000 * b + i) + rng.integers(0, 256, 400, dtype=np.uint8).tobytes() for i in range(290)]
        else:
            rows = [struct.pack("<i", 1000 * b + i) for i in range(290)]
        blk = pack_rows(L, rows, 2 if b % 2 else 1, 131072)
        assert len(blk) == 1
        m = host.COMP_ZSTD if b % 3 == 0 else host.COMP_LZ4
        fb = (C.c_uint32 * 1)(L.cryo_memrel_reserve(mem))
        assert L.cryo_stage_write_batch(C.byref(rel), blk[0], 1, m, 5, fb) == 0
        if b == 4:
            L.cryo_memrel_reserve(mem)          # a reserved-but-unwritten page in the middle (EMPTY_BLOCK)
        all_rows.append(rows)
        firsts.append(fb[0])
        methods.append(m)
    it = L.cryo_seqscan_iter_create()
    calls0 = L.cryo_cache_codec_calls()
    got_rows, got_starts = [], []
    while True:
        K = 4
        starts, ents, errs = (C.c_uint32 * K)(), (C.c_int * K)(), (C.c_int * K)()
        n = L.cryo_scan_next_batch(C.byref(rel), it, K, starts, ents, errs)
        if n == 0:
            break
        for i in range(n):
            assert errs[i] == 0
            got_starts.append(starts[i])
            got_rows.append(fetch_rows(L, L.cryo_cache_get_data(ents[i])))
    assert got_starts == firsts                       # block starts only, in increasing order
    assert got_rows == all_rows
    assert 3 <= L.cryo_cache_codec_calls() - calls0 <= 6   # 11 chains, K = 4, two methods -> at most 2 calls per batch
    assert not errors
    L.cryo_seqscan_iter_free(it)
    L.cryo_memrel_destroy(mem)


def test_c_host_batch_api_pipelined_path(codec, oracle):
    """K-block calls of 64 MiB and more are cut into chunks that overlap host copies, PCIe transfers and kernels
    (two pinned buffers each way, a second stream): same bytes, sizes and statuses as the one-shot path -- odd
    chunk tails, a padded destination stride, a corrupt block and an empty one among the inputs."""
    from pg_cryogen_amd import METHOD_LZ4, METHOD_ZSTD, bound
    L = codec.L
    B, n = 131072, 777                       # 97 MiB: 8 / 4 chunks with a ragged last one
    uniq = [oracle.synth(9, i, B, i % 5) for i in range(24)]
    raw = np.concatenate([uniq[i % 24] for i in range(n)])
    for method, param in ((METHOD_LZ4, 1), (METHOD_ZSTD, 1)):
        enc = (oracle.lz4_compress if method == METHOD_LZ4 else oracle.zstd_compress)
        exp = [enc(u, param) for u in uniq]
        stride = bound(method, B) + 4096 + 40        # padded, not a multiple of 16
        out = np.zeros(n * stride, np.uint8)
        sizes = np.zeros(n, np.uint32)
        rc = L.cryo_codec_compress_blocks(codec.h, method, param, raw.ctypes.data, B, n, out.ctypes.data, stride, sizes.ctypes.data)
        assert rc == 0
        comps = []
        for i in range(n):
            e = exp[i % 24]
            assert int(sizes[i]) == len(e) and np.array_equal(out[i * stride:i * stride + len(e)], e), (method, i)
            comps.append(np.ascontiguousarray(out[i * stride:i * stride + len(e)]))
        comps[300] = comps[300][:200].copy()         # truncated
        comps[301] = np.zeros(0, np.uint8)           # empty
        ptrs = (C.c_void_p * n)(*[c.ctypes.data if len(c) else None for c in comps])
        csz = np.array([len(c) for c in comps], np.uint32)
        dec = np.full(n * B, 0x5A, np.uint8)
        st = np.zeros(n, np.int32)
        rc = L.cryo_codec_decompress_blocks(codec.h, method, ptrs, csz.ctypes.data, n, dec.ctypes.data, B, st.ctypes.data)
        assert rc == 0
        for i in range(n):
            if i in (300, 301):
                assert st[i] != 0, i
            else:
                assert st[i] == 0 and np.array_equal(dec[i * B:(i + 1) * B], uniq[i % 24]), (method, i)


def test_workspace_options_and_trim(oracle):
    """A long-lived backend: the device workspace is bounded per call (CRYO_OPT_WORKSPACE_MAX_BYTES: the zstd pipeline
    runs fewer tiles at once), given back after host-buffer calls beyond CRYO_OPT_WORKSPACE_KEEP_BYTES, and
    cryo_codec_trim frees everything a handle holds between bursts -- bytes are the same every time."""
    import torch
    from pg_cryogen_amd import Codec, METHOD_LZ4, METHOD_ZSTD, codec as cc
    B, n = 131072, 96
    raws = [oracle.synth(12, i, B, i % 5) for i in range(n)]
    zc = [oracle.zstd_compress(r, 1 + (i % 3)) for i, r in enumerate(raws)]
    lc = [oracle.lz4_compress(r, 1) for r in raws]

    def check(c):
        for method, comps in ((METHOD_ZSTD, zc), (METHOD_LZ4, lc)):
            outs, st = c.decompress_blocks(method, comps, B)
            assert (st == 0).all() and all(np.array_equal(o, r) for o, r in zip(outs, raws))
    with Codec(0) as c:
        assert c.get_option(cc.OPT_WORKSPACE_KEEP_BYTES) == -1 and c.get_option(cc.OPT_WORKSPACE_MAX_BYTES) == 0
        check(c)
        free0 = torch.cuda.mem_get_info(0)[0]
        c.trim()
        free1 = torch.cuda.mem_get_info(0)[0]
        assert free1 > free0, "cryo_codec_trim gave nothing back"
        check(c)                                           # everything comes back with the next call
        c.set_option(cc.OPT_WORKSPACE_KEEP_BYTES, 0)       # host-buffer calls return their workspace when they end
        c.set_option(cc.OPT_LZ4_DECODE_PATH, cc.LZ4_PATH_INDEXED)
        check(c)
        check(c)
        c.set_option(cc.OPT_WORKSPACE_KEEP_BYTES, -1)
        c.set_option(cc.OPT_WORKSPACE_MAX_BYTES, 64 << 20)  # below one zstd tile of this batch: one tile in flight, still decodes
        check(c)
        c.set_option(cc.OPT_WORKSPACE_MAX_BYTES, 0)
        c.set_option(cc.OPT_LZ4_DECODE_PATH, cc.LZ4_PATH_AUTO)
        check(c)
    # the same limit through the multi-GPU dispatcher (two handles on this GPU): each handle's share of a call gives its
    # workspace AND its staging areas back when the limit says so (round 4 trimmed only single-handle calls)
    import ctypes as C
    L = cc.lib()
    h = C.c_void_p()
    assert L.cryo_multi_open((C.c_int * 2)(0, 0), 2, C.byref(h)) == 0
    try:
        def multi_check():
            for method, comps in ((METHOD_ZSTD, zc), (METHOD_LZ4, lc)):
                ptrs = (C.c_void_p * n)(*[x.ctypes.data for x in comps])
                csz = np.array([len(x) for x in comps], np.uint32)
                dec = np.zeros(n * B, np.uint8)
                st = np.ones(n, np.int32)
                assert L.cryo_multi_decompress_blocks(h, method, ptrs, csz.ctypes.data, n, dec.ctypes.data, B, st.ctypes.data) == 0
                assert (st == 0).all() and all(np.array_equal(dec[i * B:(i + 1) * B], raws[i]) for i in range(n))
        multi_check()                                          # keep = -1 (a bare handle): everything stays
        held = torch.cuda.mem_get_info(0)[0]
        assert L.cryo_multi_set_option(h, cc.OPT_WORKSPACE_KEEP_BYTES, 0) == 0
        multi_check()                                          # every share ends by giving back what it holds
        after = torch.cuda.mem_get_info(0)[0]
        assert after > held + (n // 2) * B, "the dispatcher's handles kept their staging areas (%d -> %d bytes free)" % (held, after)
    finally:
        L.cryo_multi_close(h)


def test_pipelined_pointer_forms(codec, oracle):
    """Round 4: the one-destination-per-block decode (the cache's scatter call) and a handle's share of a multi-GPU call
    take the pipelined staging too.  97 MiB calls: a destination per block in shuffled order, a corrupt and an empty
    block that must leave their destinations untouched; two handles on one GPU through cryo_multi_* (block i -> handle
    i mod 2, every share pipelined), bytes and sizes equal to the oracle's."""
    import ctypes as C
    from pg_cryogen_amd import METHOD_LZ4, METHOD_ZSTD, bound, codec as cc
    L = codec.L
    B, n = 131072, 777
    uniq = [oracle.synth(10, i, B, i % 5) for i in range(24)]
    raw = np.concatenate([uniq[i % 24] for i in range(n)])
    rng = np.random.default_rng(5)
    for method, param in ((METHOD_LZ4, 1), (METHOD_ZSTD, 1)):
        enc = (oracle.lz4_compress if method == METHOD_LZ4 else oracle.zstd_compress)
        exp = [enc(u, param) for u in uniq]
        comps = [exp[i % 24] for i in range(n)]
        comps[123] = comps[123][:150].copy()
        comps[124] = np.zeros(0, np.uint8)
        ptrs = (C.c_void_p * n)(*[c.ctypes.data if len(c) else None for c in comps])
        csz = np.array([len(c) for c in comps], np.uint32)
        # scatter: destinations in a shuffled order inside one big buffer
        perm = rng.permutation(n)
        dec = np.full(n * B, 0x5A, np.uint8)
        dsts = (C.c_void_p * n)(*[dec.ctypes.data + int(perm[i]) * B for i in range(n)])
        st = np.zeros(n, np.int32)
        assert L.cryo_codec_decompress_blocks_to(codec.h, method, ptrs, csz.ctypes.data, n, dsts, B, st.ctypes.data) == 0
        for i in range(n):
            got = dec[int(perm[i]) * B:(int(perm[i]) + 1) * B]
            if i in (123, 124):
                assert st[i] != 0 and (got == 0x5A).all(), i          # failed: destination untouched
            else:
                assert st[i] == 0 and np.array_equal(got, uniq[i % 24]), (method, i)
        # two handles behind one call
        h = C.c_void_p()
        assert L.cryo_multi_open((C.c_int * 2)(0, 0), 2, C.byref(h)) == 0
        try:
            stride = bound(method, B) + 24
            out = np.zeros(n * stride, np.uint8)
            sizes = np.zeros(n, np.uint32)
            assert L.cryo_multi_compress_blocks(h, method, param, raw.ctypes.data, B, n, out.ctypes.data, stride, sizes.ctypes.data) == 0
            for i in range(n):
                e = exp[i % 24]
                assert int(sizes[i]) == len(e) and np.array_equal(out[i * stride:i * stride + len(e)], e), (method, i)
            dec2 = np.full(n * B, 0xA5, np.uint8)
            st2 = np.zeros(n, np.int32)
            assert L.cryo_multi_decompress_blocks(h, method, ptrs, csz.ctypes.data, n, dec2.ctypes.data, B, st2.ctypes.data) == 0
            for i in range(n):
                if i in (123, 124):
                    assert st2[i] != 0 and (dec2[i * B:(i + 1) * B] == 0xA5).all(), i
                else:
                    assert st2[i] == 0 and np.array_equal(dec2[i * B:(i + 1) * B], uniq[i % 24]), (method, i)
        finally:
            L.cryo_multi_close(h)
