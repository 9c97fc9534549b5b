"""PG-free plumbing tests of the host side (compression.h mirror, block layout, page-chain
staging, block cache) -- BASELINE config 1 and the reference's own regression shapes
(sql/pg_cryogen.sql:3-13,26-28), without a GPU: the codec entry points are bound to a test
double backed by the CPU oracle (tests/codec_double.py)."""
import ctypes as C
import hashlib
import struct

import numpy as np
import pytest

from pg_cryogen_amd import host

B1M = 1 << 20


@pytest.fixture()
def H():
    import codec_double
    L = host.lib()
    dbl = codec_double.OracleCodecOps()
    L.cryo_host_set_codec_ops(C.byref(dbl.ops))
    errors = []
    handler = host.ERROR_HANDLER(lambda lvl, msg: errors.append((lvl, msg.decode())) if lvl >= 20 else None)
    L.cryo_compat_set_error_handler(handler)
    host.set_block_size(B1M)
    L.cryo_define_compression_gucs()
    L.cryo_init_cache()
    yield L, dbl, errors
    L.cryo_cache_shutdown()
    L.cryo_host_set_codec_ops(None)
    L.cryo_compat_set_error_handler(host.ERROR_HANDLER(0))
    host.set_block_size(B1M)


def heap_tuple(payload, natts):
    """23-byte HeapTupleHeader (t_hoff = 24) + user data"""
    hdr = bytearray(24)
    struct.pack_into("<H", hdr, 18, natts)
    struct.pack_into("<H", hdr, 20, 0x0800)
    hdr[22] = 24
    return bytes(hdr) + payload


def pack_rows(L, rows, natts, bs):
    """multi_insert loop of reference pg_cryogen.c:633-662: insert until -1, then start a new block"""
    blocks = []
    buf = (C.c_uint8 * bs)()
    L.cryo_init_page(buf)
    for payload in rows:
        t = heap_tuple(payload, natts)
        tb = C.create_string_buffer(t, len(t))
        ht = host.HeapTupleData(len(t), C.cast(tb, C.c_void_p))
        pos = L.cryo_storage_insert(buf, C.byref(ht))
        if pos == -1:
            blocks.append(bytes(buf))
            L.cryo_init_page(buf)
            pos = L.cryo_storage_insert(buf, C.byref(ht))
            assert pos == 1
    blocks.append(bytes(buf))
    return blocks


def fetch_rows(L, data_ptr):
    out = []
    n = L.cryo_storage_ntuples(data_ptr)
    ht = host.HeapTupleData()
    for pos in range(1, n + 1):
        L.cryo_storage_fetch(data_ptr, pos, C.byref(ht))
        out.append(C.string_at(ht.t_data, ht.t_len)[24:])
    return out


def test_pages_needed_values(H):
    L, _, _ = H
    # csize -> pages
    for csize, np_ in [(7151, 1), (8144, 1), (8145, 2), (12977, 2), (131602, 17), (1052704, 130)]:
        assert L.cryo_pages_needed(csize) == np_


def test_block_layout_matches_reference_rules(H):
    L, _, _ = H
    host.set_block_size(131072)
    rows = [struct.pack("<i", i) for i in range(1, 1001)]
    blocks = pack_rows(L, rows, 1, 131072)
    # 290 tuples per block (MaxHeapTuplesPerPage - 1, storage.c:10,33)
    assert len(blocks) == 4
    b = blocks[0]
    lower, upper = struct.unpack_from("<II", b, 0)
    assert lower == 8 + 290 * 8 and upper == 131072 - 290 * 32
    off, ln = struct.unpack_from("<II", b, 8)
    assert off == 131072 - 32 and ln == 28
    assert b[lower:upper] == bytes(upper - lower)  # zero gap


def _load(L, rows, natts, method, relid=4242, batch=8, xid=777):
    mem = L.cryo_memrel_create()
    rel = host.CryoRel()
    L.cryo_memrel_bind(mem, relid, C.byref(rel))
    bs = host.get_block_size()
    blocks = pack_rows(L, rows, natts, bs)
    firsts = []
    for i in range(0, len(blocks), batch):
        chunk = blocks[i:i + batch]
        fb = (C.c_uint32 * len(chunk))(*[L.cryo_memrel_reserve(mem) for _ in chunk])
        data = b"".join(chunk)
        rc = L.cryo_stage_write_batch(C.byref(rel), data, len(chunk), method, xid, fb)
        assert rc == 0
        firsts += list(fb)
    return mem, rel, blocks, firsts


def test_config1_copy_10k_int4_lz4(H):
    """BASELINE configs[0]: COPY a 10k-row int4 table USING pg_cryogen, lz4; then count / avg / first rows"""
    L, dbl, errors = H
    host.set_int("compression_method_guc", host.COMP_LZ4)
    rows = [struct.pack("<i", i) for i in range(1, 10001)]
    mem, rel, blocks, firsts = _load(L, rows, 1, host.COMP_LZ4)
    assert len(blocks) == 35                       # ceil(10000 / 290)
    assert dbl.compress_calls == 5                 # write-behind: 35 blocks in batches of 8
    # page-chain format of the first block (pg_cryogen.c:761-805)
    p0 = C.string_at(L.cryo_memrel_page(mem, firsts[0]), host.BLCKSZ)
    first, nxt = struct.unpack_from("<II", p0, 24)
    xid, method, csize, npages = struct.unpack_from("<IiIH", p0, 32)
    assert first == firsts[0] and xid == 777 and method == host.COMP_LZ4 and npages == L.cryo_pages_needed(csize)
    lower, upper, special = struct.unpack_from("<HHH", p0, 12)
    assert upper == 8192 and special == 8192 and lower == 48 + min(8144, csize)
    assert (nxt == host.InvalidBlockNumber) == (npages == 1)
    # seq scan through the cache, one block at a time (cache.c:244-297)
    ids = []
    for fb in firsts:
        e = C.c_int(-1)
        assert L.cryo_read_data(C.byref(rel), None, fb, C.byref(e)) == host.CRYO_ERR_SUCCESS
        assert L.cryo_cache_get_xid(e.value) == 777
        ids += [struct.unpack("<i", r)[0] for r in fetch_rows(L, L.cryo_cache_get_data(e.value))]
    assert len(ids) == 10000 and sum(ids) / len(ids) == 5000.5 and ids[:10] == list(range(1, 11))
    assert not errors
    L.cryo_memrel_destroy(mem)


def test_read_ahead_batch_uses_one_codec_call(H):
    L, dbl, _ = H
    host.set_block_size(131072)
    L.cryo_cache_configure(16)
    rows = [struct.pack("<i", i) for i in range(1, 2001)]
    mem, rel, blocks, firsts = _load(L, rows, 1, host.COMP_LZ4)
    k = len(firsts)
    assert k == 7
    before = dbl.decompress_calls
    res = (C.c_int * k)()
    errs = (C.c_int * k)()
    assert L.cryo_read_data_batch(C.byref(rel), (C.c_uint32 * k)(*firsts), k, res, errs) == 0
    assert dbl.decompress_calls == before + 1 and L.cryo_cache_codec_calls() == 1
    got = []
    for e in res:
        got += [struct.unpack("<i", r)[0] for r in fetch_rows(L, L.cryo_cache_get_data(e))]
    assert got == list(range(1, 2001))
    # second pass: all hits, no codec call
    assert L.cryo_read_data_batch(C.byref(rel), (C.c_uint32 * k)(*firsts), k, res, errs) == 0
    assert dbl.decompress_calls == before + 1 and L.cryo_cache_hits() == k
    L.cryo_memrel_destroy(mem)


def test_reference_regression_shape_zstd_then_lz4_mixed_table(H):
    """sql/pg_cryogen.sql:3-13,26-28: 500 rows (int4, md5 text) with zstd level 1, then the same
    rows again with lz4 into the same table; the method is per block (page header, cache.c:133)"""
    L, dbl, _ = H
    if dbl.stock.zstd is None:
        pytest.skip("libzstd.so.1 not loadable (the test double compresses zstd with it)")
    rows = [struct.pack("<i", i) + bytes([(32 + 1) << 1 | 1]) + hashlib.md5(str(i).encode()).hexdigest().encode()
            for i in range(1, 501)]
    mem, rel, blocks, f1 = _load(L, rows, 2, host.COMP_ZSTD)
    assert len(blocks) == 2
    bs = host.get_block_size()
    fb = (C.c_uint32 * 2)(L.cryo_memrel_reserve(mem), L.cryo_memrel_reserve(mem))
    assert L.cryo_stage_write_batch(C.byref(rel), b"".join(blocks), 2, host.COMP_LZ4, 778, fb) == 0
    total, ids = 0, []
    for b in f1 + list(fb):
        e = C.c_int(-1)
        assert L.cryo_read_data(C.byref(rel), None, b, C.byref(e)) == 0
        r = fetch_rows(L, L.cryo_cache_get_data(e.value))
        total += len(r)
        ids += [struct.unpack_from("<i", x)[0] for x in r]
    assert total == 1000 and sum(ids[:500]) / 500 == 250.5
    methods = [struct.unpack_from("<i", C.string_at(L.cryo_memrel_page(mem, b), 64), 36)[0] for b in f1 + list(fb)]
    assert methods == [1, 1, 0, 0]
    L.cryo_memrel_destroy(mem)


def test_cache_contract(H):
    L, dbl, errors = H
    host.set_block_size(4096)
    L.cryo_cache_configure(4)
    rows = [struct.pack("<i", i) for i in range(1, 601)]
    mem, rel, blocks, firsts = _load(L, rows, 1, host.COMP_LZ4)
    assert len(firsts) >= 6
    e = C.c_int(-1)
    for fb in firsts[:4]:
        assert L.cryo_read_data(C.byref(rel), None, fb, C.byref(e)) == 0
    # touching block 0 makes block 1 the least recently used
    assert L.cryo_read_data(C.byref(rel), None, firsts[0], C.byref(e)) == 0
    L.cryo_cache_get_data(e.value)
    misses = L.cryo_cache_misses()
    assert L.cryo_read_data(C.byref(rel), None, firsts[4], C.byref(e)) == 0      # evicts block 1 (true LRU)
    assert L.cryo_read_data(C.byref(rel), None, firsts[0], C.byref(e)) == 0      # still cached
    assert L.cryo_cache_misses() == misses + 1
    assert L.cryo_read_data(C.byref(rel), None, firsts[1], C.byref(e)) == 0      # reload
    assert L.cryo_cache_misses() == misses + 2
    # wrong starting block: metapage, beyond the end, and (with a multi-page chain) a continuation page
    assert L.cryo_read_data(C.byref(rel), None, 0, C.byref(e)) == host.CRYO_ERR_WRONG_STARTING_BLOCK
    assert L.cryo_read_data(C.byref(rel), None, 10 ** 6, C.byref(e)) == host.CRYO_ERR_WRONG_STARTING_BLOCK
    # reserved but never written page
    nb = L.cryo_memrel_reserve(mem)
    assert L.cryo_read_data(C.byref(rel), None, nb, C.byref(e)) == host.CRYO_ERR_EMPTY_BLOCK
    assert L.cryo_cache_err(host.CRYO_ERR_EMPTY_BLOCK) == b"empty block"
    # frozen bit in the visibility map -> FrozenTransactionId (cache.c:145-147)
    L.cryo_cache_invalidate_relation(4242)
    L.cryo_memrel_set_frozen(mem, firsts[2], True)
    assert L.cryo_read_data(C.byref(rel), None, firsts[2], C.byref(e)) == 0
    assert L.cryo_cache_get_xid(e.value) == 2
    # pinned slots (insert buffers) are never evicted; all pinned -> CACHE_IS_FULL
    L.cryo_cache_invalidate_relation(4242)
    pins = [L.cryo_cache_allocate(C.byref(rel), 1000 + i) for i in range(4)]
    assert sorted(pins) == [0, 1, 2, 3]
    assert L.cryo_read_data(C.byref(rel), None, firsts[0], C.byref(e)) == host.CRYO_ERR_CACHE_IS_FULL
    L.cryo_cache_release(pins[0])
    assert L.cryo_read_data(C.byref(rel), None, firsts[0], C.byref(e)) == 0
    assert not errors
    L.cryo_cache_release(e.value)            # releasing a read-only entry is an error (cache.c:334-335)
    assert errors and "read-only" in errors[-1][1]
    # corrupted payload -> DECOMPRESSION_FAILED (cache.c:178-179)
    L.cryo_cache_invalidate_relation(4242)
    page = L.cryo_memrel_page(mem, firsts[3])
    C.memset(page + 48, 0xFF, 64)
    assert L.cryo_read_data(C.byref(rel), None, firsts[3], C.byref(e)) == host.CRYO_ERR_DECOMPRESSION_FAILED
    L.cryo_memrel_destroy(mem)


def test_multi_page_chain_and_continuation_block(H):
    L, dbl, _ = H
    host.set_block_size(131072)
    L.cryo_cache_configure(4)
    rng = np.random.default_rng(1)
    rows = [struct.pack("<i", i) + rng.integers(0, 256, 400, dtype=np.uint8).tobytes() for i in range(1, 291)]
    mem, rel, blocks, firsts = _load(L, rows, 2, host.COMP_LZ4)
    e = C.c_int(-1)
    assert L.cryo_read_data(C.byref(rel), None, firsts[0], C.byref(e)) == 0
    npg = L.cryo_cache_get_pg_nblocks(e.value)
    assert npg >= 15            # incompressible rows: ~117 KB compressed -> 15+ pages
    assert fetch_rows(L, L.cryo_cache_get_data(e.value)) == rows
    # a continuation page is not a block start (cache.c:122-130)
    p0 = C.string_at(L.cryo_memrel_page(mem, firsts[0]), 64)
    nxt = struct.unpack_from("<I", p0, 28)[0]
    assert L.cryo_read_data(C.byref(rel), None, nxt, C.byref(e)) == host.CRYO_ERR_WRONG_STARTING_BLOCK
    L.cryo_memrel_destroy(mem)


def test_compression_h_surface(H):
    """cryo_compress / cryo_decompress keep the reference's contract (compression.c:125-159)"""
    L, dbl, errors = H
    host.set_block_size(131072)
    assert host.get_int("compression_method_guc") == host.COMP_ZSTD      # reference default, compression.c:16
    assert host.get_int("lz4_acceleration_guc") == 1 and host.get_int("zstd_compression_level_guc") == 1
    raw = dbl.ora.synth(0, 0, 131072, 1)
    n = C.c_size_t(0)
    p = L.cryo_compress(host.COMP_LZ4, raw.ctypes.data, C.byref(n))
    comp = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,)).copy()
    assert np.array_equal(comp, dbl.ora.lz4_compress(raw, 1))
    host.set_int("lz4_acceleration_guc", 50)
    p2 = L.cryo_compress(host.COMP_LZ4, raw.ctypes.data, C.byref(n))
    assert n.value == len(dbl.ora.lz4_compress(raw, 50))
    out = np.zeros(131072, np.uint8)
    assert L.cryo_decompress(host.COMP_LZ4, comp.ctypes.data, len(comp), out.ctypes.data) is True
    assert np.array_equal(out, raw)
    assert L.cryo_decompress(host.COMP_LZ4, comp.ctypes.data, len(comp) - 7, out.ctypes.data) is False
    assert not errors
    L.cryo_decompress(7, comp.ctypes.data, len(comp), out.ctypes.data)
    assert errors and errors[-1][1] == "pg_cryogen: unknown compression method"
    host.set_int("lz4_acceleration_guc", 1)


def test_scan_iterator_semantics(H):
    """reference scan_iterator.c:45-127"""
    L, _, errors = H
    it = L.cryo_seqscan_iter_create()
    assert [L.cryo_seqscan_iter_next(it) for _ in range(3)] == [1, 2, 3]     # starts at block 1 (0 = metapage)
    assert L.cryo_seqscan_iter_exclude(it, 5, False) and L.cryo_seqscan_iter_nranges(it) == 2   # split
    assert L.cryo_seqscan_iter_exclude(it, 4, False) and L.cryo_seqscan_iter_nranges(it) == 1   # range [4,4] vanishes
    assert L.cryo_seqscan_iter_next(it) == 6
    assert L.cryo_seqscan_iter_exclude(it, 7, False)                                             # start of a range
    assert L.cryo_seqscan_iter_next(it) == 8
    assert not L.cryo_seqscan_iter_exclude(it, 2, True) and not errors                           # miss_ok
    L.cryo_seqscan_iter_exclude(it, 2, False)
    assert errors and "block 2 is not the part of seqscan iterator" in errors[-1][1]
    L.cryo_seqscan_iter_reset(it)
    assert L.cryo_seqscan_iter_next(it) == 1
    L.cryo_seqscan_iter_free(it)


def test_seqscan_read_ahead_in_iterator_order(H):
    """multi-page chains interleaved with single-page ones: the read-ahead pops block starts in the
    reference's order, never offers continuation pages, and decodes K chains per codec call"""
    L, dbl, errors = H
    host.set_block_size(131072)
    L.cryo_cache_configure(32)
    rng = np.random.default_rng(2)
    mem = L.cryo_memrel_create()
    rel = host.CryoRel()
    L.cryo_memrel_bind(mem, 7, C.byref(rel))
    all_rows, firsts = [], []
    for b in range(9):
        if b % 2:   # incompressible rows -> ~15-page chain
            rows = [struct.pack("<i", 1000 * b + i) + rng.integers(0, 256, 400, dtype=np.uint8).tobytes() for i in range(290)]
        else:
            rows = [struct.pack("<i", 1000 * b + i) for i in range(290)]
        blk = pack_rows(L, rows, 2 if b % 2 else 1, 131072)
        assert len(blk) == 1
        fb = (C.c_uint32 * 1)(L.cryo_memrel_reserve(mem))
        assert L.cryo_stage_write_batch(C.byref(rel), blk[0], 1, host.COMP_LZ4, 5, fb) == 0
        if b == 4:
            L.cryo_memrel_reserve(mem)          # a reserved-but-unwritten page in the middle (EMPTY_BLOCK)
        all_rows.append(rows)
        firsts.append(fb[0])
    it = L.cryo_seqscan_iter_create()
    before = dbl.decompress_calls
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
    assert dbl.decompress_calls - before == 3         # 9 chains, K = 4 -> 3 codec calls
    assert not errors
    L.cryo_seqscan_iter_free(it)
    L.cryo_memrel_destroy(mem)


def _mixed_chain_table(L, rng, relid=7):
    """nine cryo blocks: multi-page chains interleaved with single-page ones, one reserved-but-unwritten page"""
    mem = L.cryo_memrel_create()
    rel = host.CryoRel()
    L.cryo_memrel_bind(mem, relid, C.byref(rel))
    all_rows, firsts = [], []
    for b in range(9):
        if b % 2:   # incompressible rows -> ~15-page chain
            rows = [struct.pack("<i", 1000 * b + i) + rng.integers(0, 256, 400, dtype=np.uint8).tobytes() for i in range(290)]
        else:
            rows = [struct.pack("<i", 1000 * b + i) for i in range(290)]
        blk = pack_rows(L, rows, 2 if b % 2 else 1, 131072)
        assert len(blk) == 1
        fb = (C.c_uint32 * 1)(L.cryo_memrel_reserve(mem))
        assert L.cryo_stage_write_batch(C.byref(rel), blk[0], 1, host.COMP_LZ4, 5, fb) == 0
        if b == 4:
            L.cryo_memrel_reserve(mem)          # EMPTY_BLOCK in the middle
        all_rows.append(rows)
        firsts.append(fb[0])
    return mem, rel, all_rows, firsts


def _reference_seqscan(L, rel, mem, limit=None):
    """the loop of reference pg_cryogen.c:250-275 (cryo_getnextslot): one iterator pop and one cryo_read_data per block"""
    it = L.cryo_seqscan_iter_create()
    popped, rows = [], []
    while limit is None or len(rows) < limit:
        b = L.cryo_seqscan_iter_next(it)
        if L.cryo_memrel_nblocks(mem) <= b:
            break
        popped.append(b)
        e = C.c_int(-1)
        err = L.cryo_read_data(C.byref(rel), it, b, C.byref(e))
        if err == host.CRYO_ERR_EMPTY_BLOCK:
            continue
        assert err == host.CRYO_ERR_SUCCESS, (b, err)
        rows.append(fetch_rows(L, L.cryo_cache_get_data(e.value)))
    nr = L.cryo_seqscan_iter_nranges(it)
    L.cryo_seqscan_iter_free(it)
    return popped, rows, nr


def test_unchanged_am_scan_reaches_the_batch_path(H):
    """the unmodified table AM asks for ONE block per call (reference pg_cryogen.c:262-265,
    cache.c:244-297).  A miss of a sequential scan now also loads the next block starts with the same codec call
    (pg_cryogen.gpu_readahead_blocks, never more than half the evictable slots); the scan's own iterator pops exactly
    the pages it popped before, the rows are the same, and a scan that stops early wastes at most K - 1 decodes."""
    L, dbl, errors = H
    host.set_block_size(131072)
    guc = C.c_int.in_dll(L, "cryo_gpu_readahead_blocks_guc")
    assert guc.value == 8                                   # the default
    rng = np.random.default_rng(2)
    mem, rel, all_rows, firsts = _mixed_chain_table(L, rng)
    runs = {}
    for k in (1, 4, 8):
        L.cryo_cache_configure(32)
        guc.value = k
        before = dbl.decompress_calls
        popped, rows, nr = _reference_seqscan(L, rel, mem)
        runs[k] = (popped, nr)
        assert rows == all_rows
        assert dbl.decompress_calls - before == -(-9 // k), k   # 9 chains: one codec call per K misses
    assert runs[4] == runs[1] and runs[8] == runs[1]        # iterator order and exclusions unchanged by the read-ahead
    assert [b for b in runs[1][0] if b in firsts] == firsts
    # half the evictable slots bound K: 6 slots -> 3 blocks per call
    L.cryo_cache_configure(6)
    guc.value = 8
    before = dbl.decompress_calls
    popped, rows, _ = _reference_seqscan(L, rel, mem)
    assert rows == all_rows and popped == runs[1][0]
    assert dbl.decompress_calls - before == 3
    # LIMIT 1: one codec call, at most K blocks decoded, and the look-ahead left the scan's iterator alone
    L.cryo_cache_configure(32)
    guc.value = 4
    calls0, blocks0 = dbl.decompress_calls, dbl.blocks_decompressed
    popped, rows, _ = _reference_seqscan(L, rel, mem, limit=1)
    assert rows == all_rows[:1] and popped == runs[1][0][:1]
    assert dbl.decompress_calls - calls0 == 1 and dbl.blocks_decompressed - blocks0 <= 4
    # a scan without an iterator (bitmap and tid fetches pass NULL, reference pg_cryogen.c:423,873) never reads ahead
    L.cryo_cache_configure(32)
    calls0, blocks0 = dbl.decompress_calls, dbl.blocks_decompressed
    e = C.c_int(-1)
    assert L.cryo_read_data(C.byref(rel), None, firsts[0], C.byref(e)) == host.CRYO_ERR_SUCCESS
    assert dbl.blocks_decompressed - blocks0 == 1
    guc.value = 8
    assert not errors
    L.cryo_memrel_destroy(mem)


def test_batch_larger_than_cache_never_aliases_slots(H):
    """A batch with more blocks than evictable slots: hits and earlier misses of the batch stay pinned, the
    surplus gets CACHE_IS_FULL, and every delivered entry holds its own block (round-1 advisor finding:
    a later miss evicted an earlier hit and entry 0 silently carried another block's rows)."""
    L, dbl, errors = H
    host.set_block_size(4096)
    L.cryo_cache_configure(4)
    rows = [struct.pack("<i", i) for i in range(1, 801)]
    mem, rel, blocks, firsts = _load(L, rows, 1, host.COMP_LZ4)
    assert len(firsts) >= 6
    e = C.c_int(-1)
    assert L.cryo_read_data(C.byref(rel), None, firsts[0], C.byref(e)) == 0        # block 0 cached
    want0 = fetch_rows(L, L.cryo_cache_get_data(e.value))
    pin = L.cryo_cache_allocate(C.byref(rel), 9999)                               # one slot held by an insert buffer
    k = 4
    res, errs = (C.c_int * k)(), (C.c_int * k)()
    blks = (C.c_uint32 * k)(firsts[0], firsts[1], firsts[2], firsts[3])
    rc = L.cryo_read_data_batch(C.byref(rel), blks, k, res, errs)
    assert rc == host.CRYO_ERR_CACHE_IS_FULL
    assert list(errs) == [0, 0, 0, host.CRYO_ERR_CACHE_IS_FULL] and res[3] == -1
    assert len({res[0], res[1], res[2], pin}) == 4                                 # four distinct slots
    assert fetch_rows(L, L.cryo_cache_get_data(res[0])) == want0                   # the hit still holds block 0
    got = [struct.unpack("<i", r)[0] for i in range(3) for r in fetch_rows(L, L.cryo_cache_get_data(res[i]))]
    assert got == list(range(1, len(got) + 1))                                     # blocks 0,1,2 in order, own rows
    # repeated block numbers inside one batch share a slot
    L.cryo_cache_release(pin)
    blks = (C.c_uint32 * 3)(firsts[4], firsts[4], firsts[5])
    res, errs = (C.c_int * 3)(), (C.c_int * 3)()
    assert L.cryo_read_data_batch(C.byref(rel), blks, 3, res, errs) == 0
    assert res[0] == res[1] != res[2]
    assert not errors
    L.cryo_memrel_destroy(mem)


def test_failed_probe_leaves_the_cache_untouched(H):
    """a continuation-page probe (bitmap scans do this, reference pg_cryogen.c:442-444) fails with
    WRONG_STARTING_BLOCK and must not clobber the LRU victim's page list or xid (round-1 advisor finding)"""
    L, dbl, errors = H
    host.set_block_size(131072)
    L.cryo_cache_configure(1)
    rng = np.random.default_rng(3)
    rows = [struct.pack("<i", i) + rng.integers(0, 256, 400, dtype=np.uint8).tobytes() for i in range(1, 291)]
    mem, rel, blocks, firsts = _load(L, rows, 2, host.COMP_LZ4)
    e = C.c_int(-1)
    assert L.cryo_read_data(C.byref(rel), None, firsts[0], C.byref(e)) == 0
    npg, xid = L.cryo_cache_get_pg_nblocks(e.value), L.cryo_cache_get_xid(e.value)
    assert npg >= 15
    nxt = struct.unpack_from("<I", C.string_at(L.cryo_memrel_page(mem, firsts[0]), 64), 28)[0]
    e2 = C.c_int(-1)
    assert L.cryo_read_data(C.byref(rel), None, nxt, C.byref(e2)) == host.CRYO_ERR_WRONG_STARTING_BLOCK
    misses = L.cryo_cache_misses()
    assert L.cryo_read_data(C.byref(rel), None, firsts[0], C.byref(e)) == 0        # still a hit, metadata intact
    assert L.cryo_cache_misses() == misses
    assert L.cryo_cache_get_pg_nblocks(e.value) == npg and L.cryo_cache_get_xid(e.value) == xid
    assert fetch_rows(L, L.cryo_cache_get_data(e.value)) == rows
    L.cryo_memrel_destroy(mem)


def test_invalidation_callback_never_opens_the_codec():
    """cryo_cache_invalidate_relation is PostgreSQL's relcache callback (reference pg_cryogen.c:163-167): it fires for
    every invalidation of any relation in every backend that loaded the extension.  It must use the codec binding only
    if the backend has one already -- in a fresh process with no GPU the lazy open would fail and leave its message in
    cryo_host_codec_error(); nothing may have tried."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys, ctypes as C
        sys.path.insert(0, %r)
        os.environ.pop("CRYO_HOST_TEST_HOOKS", None)
        from pg_cryogen_amd import host
        host.use(production=True)
        L = host.lib()
        L.cryo_host_codec_error.restype = C.c_char_p
        L.cryo_host_codec_ops_if_open.restype = C.c_void_p
        L.cryo_cache_configure(4)
        L.cryo_cache_invalidate_relation(1234)
        L.cryo_cache_invalidate_relation(0)
        assert L.cryo_host_codec_error() == b"", L.cryo_host_codec_error()
        assert not L.cryo_host_codec_ops_if_open()
        print("ok")
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
