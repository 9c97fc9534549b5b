"""GPU tests of the device-resident block pool ("optional device-resident compressed/decompressed
pool so repeated scans skip PCIe"; the reference's cache is host memory only, cache.c:17-50).

Through the C ABI (cryo_codec_decompress_blocks_keyed) and through the shipped host library (cryo_read_data_batch
with pg_cryogen.gpu_pool_mb): a second read of the same blocks moves ZERO bytes towards the device and returns
the same bytes."""
import ctypes as C
import struct

import numpy as np
import pytest

from pg_cryogen_amd import METHOD_LZ4, METHOD_ZSTD, host
from pg_cryogen_amd import codec as cc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def pooled(codec):
    codec.set_option(cc.OPT_POOL_BYTES, 0)
    yield codec
    codec.set_option(cc.OPT_POOL_BYTES, 0)


def test_pool_second_call_moves_nothing_to_the_device(pooled, oracle):
    c = pooled
    B, n = 131072, 24
    raws = [oracle.synth(41, i, B, i % 5) for i in range(n)]
    for method, comp in ((METHOD_LZ4, lambda r: oracle.lz4_compress(r, 1)), (METHOD_ZSTD, lambda r: oracle.zstd_compress(r, 1))):
        comps = [comp(r) for r in raws]
        keys = [(7 << 32) | (100 + i) for i in range(n)]
        c.set_option(cc.OPT_POOL_BYTES, 64 * B)
        c.pool_invalidate(everything=True)
        t0 = c.transfer_counters()
        outs, st = c.decompress_blocks_keyed(method, keys, comps, B)
        t1 = c.transfer_counters()
        assert (st == 0).all() and all(np.array_equal(o, r) for o, r in zip(outs, raws))
        assert t1["pool_misses"] - t0["pool_misses"] == n and t1["h2d_bytes"] > t0["h2d_bytes"]
        assert t1["pool_blocks"] == n and t1["pool_capacity"] == 64
        outs, st = c.decompress_blocks_keyed(method, keys, comps, B)
        t2 = c.transfer_counters()
        assert (st == 0).all() and all(np.array_equal(o, r) for o, r in zip(outs, raws))
        assert t2["pool_hits"] - t1["pool_hits"] == n
        assert t2["h2d_bytes"] == t1["h2d_bytes"], "a pooled re-read moved bytes towards the device"
        # a mix: some known, some new, one unkeyed (never kept), order scrambled
        more = [oracle.synth(42, i, B, 1) for i in range(5)]
        mk = keys[3:9] + [(7 << 32) | (900 + i) for i in range(4)] + [0]
        mc = comps[3:9] + [comp(r) for r in more]
        outs, st = c.decompress_blocks_keyed(method, mk, mc, B)
        t3 = c.transfer_counters()
        assert (st == 0).all()
        for o, r in zip(outs, raws[3:9] + more):
            assert np.array_equal(o, r)
        assert t3["pool_hits"] - t2["pool_hits"] == 6 and t3["pool_misses"] - t2["pool_misses"] == 5
        # the relation is invalidated (reference: relcache callback pg_cryogen.c:163-167): its blocks are decoded again
        c.pool_invalidate(key_hi=7)
        outs, st = c.decompress_blocks_keyed(method, keys[:4], comps[:4], B)
        t4 = c.transfer_counters()
        assert t4["pool_hits"] == t3["pool_hits"] and t4["pool_misses"] - t3["pool_misses"] == 4
        assert all(np.array_equal(o, r) for o, r in zip(outs, raws[:4]))


def test_pool_wraps_replaces_and_never_keeps_bad_blocks(pooled, oracle):
    c = pooled
    B = 131072
    c.set_option(cc.OPT_POOL_BYTES, 4 * B)            # four slots, first in first out
    c.pool_invalidate(everything=True)
    raws = [oracle.synth(43, i, B, i % 3) for i in range(10)]
    comps = [oracle.lz4_compress(r, 1) for r in raws]
    keys = [(9 << 32) | i for i in range(10)]
    outs, st = c.decompress_blocks_keyed(METHOD_LZ4, keys, comps, B)          # 10 blocks through 4 slots: three rounds
    assert (st == 0).all() and all(np.array_equal(o, r) for o, r in zip(outs, raws))
    t = c.transfer_counters()
    assert t["pool_blocks"] <= 4
    # the same key with other content (the block was rewritten): size / fingerprint differ, it is decoded again
    other = oracle.lz4_compress(oracle.synth(44, 0, B, 0), 1)
    h0 = c.transfer_counters()["pool_hits"]
    outs, st = c.decompress_blocks_keyed(METHOD_LZ4, [keys[9]], [other], B)
    assert st[0] == 0 and np.array_equal(outs[0], oracle.synth(44, 0, B, 0))
    assert c.transfer_counters()["pool_hits"] == h0
    # a rewrite of the SAME compressed size that shares the stream's first, middle and last 8 bytes (what round 3's
    # 24-byte fingerprint sampled) is a different stream: the whole stream is hashed now, it is decoded again
    base = oracle.synth(45, 0, B, 3)                                      # incompressible: stored as one literal run
    twin = base.copy()
    twin[B // 3] ^= 0x5A
    cb, ct = oracle.lz4_compress(base, 1), oracle.lz4_compress(twin, 1)
    n_ = len(cb)
    assert len(ct) == n_ and np.array_equal(cb[:8], ct[:8]) and np.array_equal(cb[n_ // 2 - 4:n_ // 2 + 4], ct[n_ // 2 - 4:n_ // 2 + 4]) \
        and np.array_equal(cb[-8:], ct[-8:]) and not np.array_equal(cb, ct)
    kk = (9 << 32) | 555
    outs, st = c.decompress_blocks_keyed(METHOD_LZ4, [kk], [cb], B)
    assert st[0] == 0 and np.array_equal(outs[0], base)
    h1 = c.transfer_counters()["pool_hits"]
    outs, st = c.decompress_blocks_keyed(METHOD_LZ4, [kk], [ct], B)
    assert st[0] == 0 and np.array_equal(outs[0], twin) and c.transfer_counters()["pool_hits"] == h1
    # a corrupt stream is reported, leaves its destination alone and is not kept
    bad = comps[2][:len(comps[2]) // 2].copy()
    outs, st = c.decompress_blocks_keyed(METHOD_LZ4, [(9 << 32) | 77, keys[9]], [bad, other], B)
    assert st[0] != 0 and outs[0] is None and st[1] == 0
    m0 = c.transfer_counters()["pool_misses"]
    outs, st = c.decompress_blocks_keyed(METHOD_LZ4, [(9 << 32) | 77], [bad], B)
    assert st[0] != 0 and c.transfer_counters()["pool_misses"] == m0 + 1
    # pool off: the keyed call is the plain scatter call
    c.set_option(cc.OPT_POOL_BYTES, 0)
    outs, st = c.decompress_blocks_keyed(METHOD_LZ4, keys[:3], comps[:3], B)
    assert (st == 0).all() and all(np.array_equal(o, r) for o, r in zip(outs, raws[:3]))
    assert c.transfer_counters()["pool_capacity"] == 0


def test_host_cache_rescan_served_from_the_device_pool(oracle):
    """the shipped host library: cryo_read_data_batch over 35 blocks twice, the host cache emptied in between --
    the second scan is served from HBM (zero bytes towards the device), rows identical; invalidating the relation
    (cryo_cache_invalidate_relation, the relcache callback of reference pg_cryogen.c:163-167) makes the third decode"""
    from test_host_plumbing import _load, fetch_rows
    host.use(production=True)
    L = host.lib()
    try:
        errors = []
        handler = host.ERROR_HANDLER(lambda lvl, msg: errors.append((lvl, msg.decode())) if lvl >= 20 else None)
        L.cryo_compat_set_error_handler(handler)
        host.set_block_size(131072)
        L.cryo_define_compression_gucs()
        host.set_int("cryo_gpu_pool_mb_guc", 64)
        L.cryo_cache_configure(40)
        rows = [struct.pack("<i", i) for i in range(1, 10001)]
        mem, rel, blocks, firsts = _load(L, rows, 1, host.COMP_LZ4, batch=16)
        k = len(firsts)
        assert k == 35 and not errors

        def scan():
            res, errs = (C.c_int * k)(), (C.c_int * k)()
            assert L.cryo_read_data_batch(C.byref(rel), (C.c_uint32 * k)(*firsts), k, res, errs) == 0
            ids = []
            for e in res:
                ids += [struct.unpack("<i", r)[0] for r in fetch_rows(L, L.cryo_cache_get_data(e))]
            return ids
        L.cryo_cache_configure(40)
        a = host.transfer_counters()
        ids1 = scan()
        b = host.transfer_counters()
        assert ids1 == list(range(1, 10001))
        assert b[3] - a[3] == k and b[0] > a[0]                       # 35 decoded, bytes went to the device
        L.cryo_cache_configure(40)                                    # the host cache forgets everything
        ids2 = scan()
        c_ = host.transfer_counters()
        assert ids2 == ids1
        assert c_[2] - b[2] == k and c_[0] == b[0], "re-scan was not served from the device pool"
        L.cryo_cache_invalidate_relation(rel.relid)
        L.cryo_cache_configure(40)
        ids3 = scan()
        d = host.transfer_counters()
        assert ids3 == ids1 and d[3] - c_[3] == k and d[0] > c_[0]
        # InvalidOid: what PostgreSQL's relcache callback passes after a sinval-queue reset ("anything may have changed"):
        # every entry of the pool goes (round 3 forwarded it as "relation 0" and dropped nothing)
        L.cryo_cache_configure(40)
        ids4 = scan()
        e_ = host.transfer_counters()
        assert ids4 == ids1 and e_[2] - d[2] == k and e_[0] == d[0]   # pool hits again
        L.cryo_cache_invalidate_relation(0)
        L.cryo_cache_configure(40)
        ids5 = scan()
        f_ = host.transfer_counters()
        assert ids5 == ids1 and f_[3] - e_[3] == k and f_[0] > e_[0], "InvalidOid did not empty the device pool"
        L.cryo_memrel_destroy(mem)
    finally:
        host.set_int("cryo_gpu_pool_mb_guc", 0)
        L.cryo_cache_shutdown()
        L.cryo_compat_set_error_handler(host.ERROR_HANDLER(0))
        host.set_block_size(1 << 20)
        host.use(production=None)
