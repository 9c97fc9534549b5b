"""GPU parity tests for the LZ4 path: HIP kernels through the C ABI vs the CPU oracle.

Reference call shapes: compression.c:70-72 (LZ4_compress_fast) and :84 (LZ4_decompress_safe).
Bar: bit-exact (integer/byte work).
"""
import numpy as np
import pytest

from pg_cryogen_amd import METHOD_LZ4

pytestmark = pytest.mark.gpu

SIZES = [131072, 1 << 20, 4096, 65546, 65547]


@pytest.mark.parametrize("B", [131072, 1 << 20, 4096])
def test_synth_matches_oracle(codec, oracle, B):
    n = 3
    for dist in range(5):
        d = codec.alloc(n * B)
        codec.synth_batch(7, 5, n, B, dist, d)
        codec.sync()
        got = d.download()
        d.free()
        for i in range(n):
            exp = oracle.synth(7, 5 + i, B, dist)
            assert np.array_equal(got[i * B:(i + 1) * B], exp), (B, dist, i)


@pytest.mark.parametrize("B", SIZES)
def test_lz4_encode_bit_exact(codec, oracle, B):
    blocks, params = [], []
    for dist in range(5):
        for blk in range(2):
            blocks.append(oracle.synth(0, blk, B, dist))
    for accel in (0, 1, 2, 7, 50):
        got = codec.compress_blocks(METHOD_LZ4, accel, blocks)
        for i, b in enumerate(blocks):
            exp = oracle.lz4_compress(b, accel)
            assert len(got[i]) == len(exp), (B, accel, i, len(got[i]), len(exp))
            assert np.array_equal(got[i], exp), (B, accel, i)


@pytest.mark.parametrize("B", SIZES)
def test_lz4_decode_bit_exact(codec, oracle, B):
    blocks, comps = [], []
    for dist in range(5):
        for accel in (1, 50):
            b = oracle.synth(1, dist, B, dist)
            blocks.append(b)
            comps.append(oracle.lz4_compress(b, accel))
    outs, st = codec.decompress_blocks(METHOD_LZ4, comps, B)
    assert (st == 0).all(), st
    for i, b in enumerate(blocks):
        assert np.array_equal(outs[i], b), (B, i)


def test_lz4_decode_rejects_like_oracle(codec, oracle):
    """Corrupted streams: success iff the oracle decodes exactly B bytes, and then equal bytes."""
    B = 4096
    rng = np.random.default_rng(3)
    comps, expect = [], []
    for dist in (0, 1, 3):
        c = oracle.lz4_compress(oracle.synth(0, 0, B, dist), 1)
        for it in range(400):
            m = c.copy()
            k = it % 4
            if k == 0:
                for _ in range(int(rng.integers(1, 4))):
                    m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
            elif k == 1:
                m = m[:int(rng.integers(1, len(m)))].copy()
            elif k == 2:
                m = np.concatenate([m, rng.integers(0, 256, int(rng.integers(1, 20))).astype(np.uint8)])
            else:
                p = int(rng.integers(0, len(m) - 1))
                m[p] = 0
                m[p + 1] = 0
            r, out = oracle.lz4_decompress(m, B, fill=0xA5)
            comps.append(m)
            expect.append(out.copy() if r == B else None)
    outs, st = codec.decompress_blocks(METHOD_LZ4, comps, B)
    n_ok = 0
    for i, e in enumerate(expect):
        if e is None:
            assert st[i] != 0, i
        else:
            n_ok += 1
            assert st[i] == 0, i
            assert np.array_equal(outs[i], e), i
    assert n_ok > 50


def test_lz4_single_block_host_api(codec, oracle):
    B = 131072
    b = oracle.synth(2, 9, B, 1)
    c = codec.compress_block(METHOD_LZ4, 1, b)
    assert np.array_equal(c, oracle.lz4_compress(b, 1))
    out = codec.decompress_block(METHOD_LZ4, c, B)
    assert np.array_equal(out, b)
    assert codec.decompress_block(METHOD_LZ4, c[:-3], B) is None


def test_lz4_decode_unaligned_inputs_and_edge_statuses(codec, oracle):
    """compressed blocks packed at odd byte offsets; empty / absurd inputs are reported, never crash"""
    from pg_cryogen_amd import METHOD_LZ4
    B = 131072
    blocks = [oracle.synth(11, i, B, i % 5) for i in range(7)]
    comps = [oracle.lz4_compress(b, 1) for b in blocks]
    comps.append(np.zeros(0, np.uint8))                       # empty stream -> corrupt
    comps.append(np.array([0x00], np.uint8))                  # valid stream of an EMPTY block -> wrong size
    n = len(comps)
    offs, pos = [], 3
    for c in comps:
        offs.append(pos)
        pos += len(c) + 1 + (len(c) % 7)                       # odd, irregular gaps
    packed = np.full(pos + 64, 0xEE, np.uint8)
    for o, c in zip(offs, comps):
        packed[o:o + len(c)] = c
    d_src, d_off, d_sz = codec.alloc(packed.nbytes), codec.alloc(8 * n), codec.alloc(4 * n)
    d_dst, d_st = codec.alloc(n * (B + 5)), codec.alloc(4 * n)
    d_src.upload(packed)
    d_off.upload(np.array(offs, np.uint64))
    d_sz.upload(np.array([len(c) for c in comps], np.uint32))
    d_dst.memset(0x5A)
    codec.decompress_batch(METHOD_LZ4, d_src, d_off, d_sz, d_dst, B + 5, B, n, d_st)   # odd dst stride too
    codec.sync()
    st = d_st.download(dtype=np.int32)
    raw = d_dst.download()
    for i, b in enumerate(blocks):
        assert st[i] == 0
        assert np.array_equal(raw[i * (B + 5):i * (B + 5) + B], b), i
        assert (raw[i * (B + 5) + B:(i + 1) * (B + 5)] == 0x5A).all()          # nothing written past the block
    assert st[7] != 0 and st[8] != 0
    for x in (d_src, d_off, d_sz, d_dst, d_st):
        x.free()


def test_lz4_checksum_of_checksums_property(codec, oracle):
    """size-independent property at a larger batch: per-block checksums of decode(encode(x)) equal those of x"""
    from pg_cryogen_amd import METHOD_LZ4, bound
    B, n = 131072, 2048
    stride = (bound(METHOD_LZ4, B) + 15) & ~15
    d_raw, d_comp, d_out = codec.alloc(n * B), codec.alloc(n * stride), codec.alloc(n * B)
    d_sz, d_st, d_off = codec.alloc(4 * n), codec.alloc(4 * n), codec.alloc(8 * n)
    d_s1, d_s2 = codec.alloc(8 * n), codec.alloc(8 * n)
    for dist in (0, 1):
        codec.synth_batch(5, 1000, n, B, dist, d_raw)
        codec.compress_batch(METHOD_LZ4, 1, d_raw, B, B, n, d_comp, stride, d_sz, d_st)
        d_off.upload(np.arange(n, dtype=np.uint64) * np.uint64(stride))
        codec.decompress_batch(METHOD_LZ4, d_comp, d_off, d_sz, d_out, B, B, n, d_st)
        codec.checksum_batch(d_raw, B, n, d_s1, fixed_size=B)
        codec.checksum_batch(d_out, B, n, d_s2, fixed_size=B)
        codec.sync()
        assert (d_st.download(dtype=np.int32) == 0).all()
        s1, s2 = d_s1.download(dtype=np.uint64), d_s2.download(dtype=np.uint64)
        assert np.array_equal(s1, s2)
        # the device checksum equals the host reference of the C ABI on a sampled block
        from pg_cryogen_amd import checksum64
        assert int(s1[17]) == checksum64(oracle.synth(5, 1017, B, dist))
    for x in (d_raw, d_comp, d_out, d_sz, d_st, d_off, d_s1, d_s2):
        x.free()


def test_lz4_encode_batch_kernel_corners(codec, oracle):
    """the 64-probes-per-step encoder (blocks >= 65547 bytes): odd sizes around its lower bound, data whose
    search steps outgrow the LDS ring (incompressible, large acceleration), far matches (beyond the ring,
    up to and past the 65535 limit), long runs, and mixtures"""
    rng = np.random.default_rng(11)
    blocks = []
    for n in (65547, 65548, 70001, 131072 + 13, 262144 + 1):
        blocks.append(rng.integers(0, 256, n, dtype=np.uint8))                       # incompressible
        a = rng.integers(0, 256, n, dtype=np.uint8)
        chunk = a[:3000].copy()
        for off in (4000, 4000 + 2047, 40000, 40000 + 65535, 40000 + 65536, n - 3100):  # repeats at many distances
            if off + 3000 <= n:
                a[off:off + 3000] = chunk
        blocks.append(a)
        z = np.zeros(n, np.uint8)
        z[n // 3:n // 3 + 100] = rng.integers(0, 256, 100, dtype=np.uint8)
        blocks.append(z)                                                               # very long matches
        t = np.frombuffer((b"abcdefghij" * (n // 10 + 1))[:n], np.uint8).copy()
        t[::997] = rng.integers(0, 256, len(t[::997]), dtype=np.uint8)
        blocks.append(t)                                                               # periodic with noise
    by_len = {}
    for b in blocks:
        by_len.setdefault(len(b), []).append(b)
    for n, lst in by_len.items():
        for accel in (1, 3, 50, 5000, 65537):
            got = codec.compress_blocks(METHOD_LZ4, accel, lst)
            for i, b in enumerate(lst):
                exp = oracle.lz4_compress(b, accel)
                assert np.array_equal(got[i], exp), (n, accel, i, len(got[i]), len(exp))


def test_lz4_encode_last_match_runs_into_the_block_end(codec, oracle):
    """the last five bytes of a block are literals (liblz4's matchlimit): a match whose bytes go on being equal up to
    the block's end stops there -- for every distance of the last match's start from the end (the encoder knows the
    first twelve bytes of a match from the search itself: matches that end inside them, at them and behind them), near
    and far candidates, with and without literals in front"""
    rng = np.random.default_rng(29)
    words = [rng.integers(0, 256, int(rng.integers(5, 13)), dtype=np.uint8) for _ in range(300)]
    for n in (65547, 70000, 131072):
        blocks = []
        for tail in range(6, 40):
            for dist in (8, 300, 5000):
                # text of 300 words: matches all along, so the search is still taking every position when it gets to the end
                a = np.concatenate([words[int(k)] for k in rng.integers(0, 300, n // 5)])[:n].copy()
                a[n - tail:] = a[n - tail - dist:n - dist]    # the block's last `tail` bytes repeat what lies `dist` back
                blocks.append(a)
                b = a.copy()
                b[n - tail - 1] ^= 0x55                        # ... with a mismatch right in front of them
                blocks.append(b)
        for accel in (1, 4):
            got = codec.compress_blocks(METHOD_LZ4, accel, blocks)
            for i, b in enumerate(blocks):
                exp = oracle.lz4_compress(b, accel)
                assert np.array_equal(got[i], exp), (n, accel, i, len(got[i]), len(exp))


def test_lz4_golden_cells_on_gpu(codec, oracle):
    """tests/golden/vectors.json (liblz4 1.9.3 called as reference compression.c:70-72): the device encoder's
    bytes hash to the golden comp_sha256 for every LZ4 cell, all sizes incl. 128 KiB and 1 MiB, and the device
    decoder turns them back into blocks that hash to raw_sha256 -- GPU against the library directly, not through
    the oracle."""
    import hashlib
    import json
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    cells = [c for c in json.load(open(os.path.join(G, "vectors.json")))["cells"] if c["method"] == "lz4"]
    assert len(cells) >= 200
    for B in sorted(set(c["B"] for c in cells)):
        for accel in sorted(set(c["param"] for c in cells if c["B"] == B)):
            sub = [c for c in cells if c["B"] == B and c["param"] == accel]
            blocks = [oracle.synth(0, c["block"], B, c["dist"]) for c in sub]
            got = codec.compress_blocks(METHOD_LZ4, accel, blocks)
            for c, b, g in zip(sub, blocks, got):
                assert sha(b) == c["raw_sha256"], c
                assert len(g) == c["csize"] and sha(g) == c["comp_sha256"], c
            outs, st = codec.decompress_blocks(METHOD_LZ4, got, B)
            assert (st == 0).all()
            for c, o in zip(sub, outs):
                assert sha(o) == c["raw_sha256"], c


def test_lz4_indexed_decoder_large_batch(codec, oracle):
    """Batches of 24576 blocks and more take the sequence-index pass + the decoder built on it (lz4_dec2.hip).
    32768 small blocks: every distribution, three accelerations, block sizes whose sequences end in every way
    (4 KiB, and 1 KiB where whole blocks are a handful of sequences), plus mutated streams among them: statuses
    and bytes must equal the oracle's, block by block."""
    from pg_cryogen_amd import METHOD_LZ4
    rng = np.random.default_rng(21)
    for B, n in ((4096, 32768), (1024, 40000)):
        uniq_raw, uniq_comp = [], []
        for dist in range(5):
            for blk in range(6):
                for accel in (1, 7, 50):
                    r = oracle.synth(3, blk, B, dist)
                    uniq_raw.append(r)
                    uniq_comp.append(oracle.lz4_compress(r, accel))
        # structured random blocks (repeats at many distances, runs, noise)
        for k in range(40):
            a = rng.integers(0, 256, B, dtype=np.uint8)
            for _ in range(int(rng.integers(1, 12))):
                ln = int(rng.integers(4, 300)); src = int(rng.integers(0, B - ln)); dst = int(rng.integers(0, B - ln))
                a[dst:dst + ln] = a[src:src + ln].copy()
            if k % 3 == 0:
                p = int(rng.integers(0, B - 600)); a[p:p + 600] = a[p]
            uniq_raw.append(a)
            uniq_comp.append(oracle.lz4_compress(a, 1))
        # run-heavy blocks: many short matches that overlap themselves (offset 1..8 < length <= ~80), at every phase
        # of a batch and across the 64-byte chunks of match space: the copy engine keeps those up to 64 bytes in the batch
        for k in range(40):
            a = np.empty(B, np.uint8)
            pos_ = 0
            while pos_ < B:
                per = int(rng.integers(1, 9)); ln = int(rng.integers(5, 80)); lit = int(rng.integers(0, 6))
                seg = np.concatenate([rng.integers(0, 256, lit + per, dtype=np.uint8), np.zeros(ln, np.uint8)])
                for j in range(lit + per, len(seg)):
                    seg[j] = seg[j - per]
                take = min(len(seg), B - pos_)
                a[pos_:pos_ + take] = seg[:take]
                pos_ += take
            uniq_raw.append(a)
            uniq_comp.append(oracle.lz4_compress(a, 1))
        # mutated streams: verdict and bytes as the oracle's
        muts = []
        for c in uniq_comp[:30]:
            for it in range(6):
                m = c.copy()
                if it % 3 == 0:
                    m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
                elif it % 3 == 1:
                    m = m[:int(rng.integers(1, len(m)))].copy()
                else:
                    p = int(rng.integers(0, len(m) - 1)); m[p] = 0; m[p + 1] = 0
                r, out = oracle.lz4_decompress(m, B, fill=0xA5)
                muts.append((m, out.copy() if r == B else None))
        nu = len(uniq_comp)
        pick = rng.integers(0, nu + len(muts), n)
        comps = [uniq_comp[i] if i < nu else muts[i - nu][0] for i in pick]
        offs = np.zeros(n, np.uint64)
        pos = 5
        for i, c in enumerate(comps):
            offs[i] = pos
            pos += len(c) + (i % 3)                      # packed at arbitrary byte offsets
        packed = np.zeros(pos + 64, np.uint8)
        for o, c in zip(offs, comps):
            packed[int(o):int(o) + len(c)] = c
        d_src, d_off, d_sz = codec.alloc(packed.nbytes), codec.alloc(8 * n), codec.alloc(4 * n)
        d_dst, d_st = codec.alloc(n * B), codec.alloc(4 * n)
        d_src.upload(packed)
        d_off.upload(offs)
        d_sz.upload(np.array([len(c) for c in comps], np.uint32))
        d_dst.memset(0xA5)
        codec.decompress_batch(METHOD_LZ4, d_src, d_off, d_sz, d_dst, B, B, n, d_st)
        codec.sync()
        st = d_st.download(dtype=np.int32)
        raw = d_dst.download().reshape(n, B)
        bad = 0
        for j, i in enumerate(pick):
            exp = uniq_raw[i] if i < nu else muts[i - nu][1]
            if exp is None:
                assert st[j] != 0, (B, j)
                bad += 1
            else:
                assert st[j] == 0, (B, j, int(i))
                assert np.array_equal(raw[j], exp), (B, j, int(i))
        assert bad > 100
        for x in (d_src, d_off, d_sz, d_dst, d_st):
            x.free()


# ---------------------------------------------------------------------------------------------------------------
# The indexed decoder (k_lz4_index + k_lz4_dec_seq: the kernels BASELINE configs[1] is measured on) at the block sizes
# it is measured on.  A per-handle option forces the path whatever the batch size (CRYO_OPT_LZ4_DECODE_PATH) and the
# number of index walkers per block (CRYO_OPT_LZ4_INDEX_WALKERS): 1 = the headline configuration (one lane per
# block), 4 / 8 = what mid-sized batches and 1 MiB blocks get, 64 = a handful of blocks.  Every decoded block is
# compared with the oracle's block, every verdict with the oracle's verdict (reference call shape: compression.c:84).
# ---------------------------------------------------------------------------------------------------------------
def _indexed(codec, walkers):
    from pg_cryogen_amd import codec as cc

    class _Ctx:
        def __enter__(self_):
            codec.set_option(cc.OPT_LZ4_DECODE_PATH, cc.LZ4_PATH_INDEXED)
            codec.set_option(cc.OPT_LZ4_INDEX_WALKERS, walkers)

        def __exit__(self_, *a):
            codec.set_option(cc.OPT_LZ4_DECODE_PATH, cc.LZ4_PATH_AUTO)
            codec.set_option(cc.OPT_LZ4_INDEX_WALKERS, 0)
    return _Ctx()


def _decode_check(codec, comps, expect, B, tag):
    """comps decoded as one device batch; expect[i] = the oracle's block, or None for 'must be rejected'.  The indexed
    decoder has two forms (CRYO_OPT_LZ4_DECODE_WAVES: one wave per block, k_lz4_dec_seq -- what big batches get; two waves
    per block, k_lz4_dec_dual -- what batches up to 3 072 blocks get): the batch goes through both."""
    from pg_cryogen_amd import codec as cc
    for waves in (1, 2):
        codec.set_option(cc.OPT_LZ4_DECODE_WAVES, waves)
        try:
            outs, st = codec.decompress_blocks(METHOD_LZ4, comps, B)
        finally:
            codec.set_option(cc.OPT_LZ4_DECODE_WAVES, 0)
        for i, e in enumerate(expect):
            if e is None:
                assert st[i] != 0, (tag, waves, i)
            else:
                assert st[i] == 0, (tag, waves, i, int(st[i]))
                assert np.array_equal(outs[i], e), (tag, waves, i)


WALKERS = [1, 4, 64]


@pytest.mark.parametrize("B", [131072, 1 << 20])
@pytest.mark.parametrize("walkers", WALKERS)
def test_indexed_decoder_distributions_at_headline_sizes(codec, oracle, B, walkers):
    """all five synthetic distributions x accelerations 1, 7, 50 (configs[1] is `wide` at acceleration 1)"""
    blocks, comps = [], []
    for dist in range(5):
        for accel in (1, 7, 50):
            for blk in range(2 if B == 131072 else 1):
                b = oracle.synth(17, 100 * dist + blk, B, dist)
                blocks.append(b)
                comps.append(oracle.lz4_compress(b, accel))
    with _indexed(codec, walkers):
        _decode_check(codec, comps, blocks, B, ("dist", B, walkers))


@pytest.mark.parametrize("walkers", [4, 16, 64])
def test_indexed_decoder_long_runs_across_segments(codec, oracle, walkers):
    """Round 5: a chain may jump over whole index segments -- the match-length bytes of one long run (the zero gap of a cryo
    block: 4 100 bytes of 255 for 1 MiB) span several 1 KiB segments, guessed starts fall inside them, and the walkers on the
    true chain are found by a sweep (k_lz4_index phase 2, k_lz4_few_path).  Blocks of text, one run of 60 KB ... 900 KB at
    varying places, text again; and two runs; through 4 / 16 / 64 walkers per block, both decoder forms, and the few-blocks
    path (up to 1 024 walkers)."""
    B = 1 << 20
    text = oracle.synth(91, 0, B, 0)
    rng = np.random.default_rng(91 + walkers)
    blocks = []
    for pre, run in ((1000, 900000), (30000, 500000), (200000, 60000), (517, 300000), (70001, 777777), (400000, 250000)):
        b = text.copy()
        b[pre:pre + run] = int(rng.integers(0, 256))
        blocks.append(b)
    b = text.copy()
    b[5000:205000] = 0
    b[600000:900000] = 0xFF                                            # a run of 255 DATA bytes: literal-free match, 255s in the match-length bytes only
    blocks.append(b)
    blocks.append(oracle.synth(91, 1, B, 1))                           # narrow rows
    blocks.append(oracle.synth(91, 2, B, 2))                           # int4 rows: periodic stream, one walker (below 16 KiB)
    comps = [oracle.lz4_compress(x, 1) for x in blocks]
    with _indexed(codec, walkers):
        _decode_check(codec, comps, blocks, B, ("runs", walkers))
    with _few_blocks(codec):
        _decode_check(codec, comps, blocks, B, ("runs-few", walkers))
        _decode_check(codec, comps[1:2], blocks[1:2], B, ("runs-few-one", walkers))


def test_indexed_decoder_two_walkers_on_1mib_blocks(codec, oracle):
    """Two walkers per 1 MiB block: a segment holds up to 81 984 records, more than the 16 bits in which a hand-over's skip
    count travels (lz4_index.hip: such a hand-over counts as a boundary that did not meet and the block is walked again by
    one walker) -- dense sequences (`wide`: ~51 000 per block), literal-heavy and highly compressible blocks alike."""
    B = 1 << 20
    blocks, comps = [], []
    for dist in range(5):
        for accel in (1, 50):
            b = oracle.synth(23, 10 * dist + accel, B, dist)
            blocks.append(b)
            comps.append(oracle.lz4_compress(b, accel))
    rng = np.random.default_rng(77)
    a = rng.integers(0, 256, B, dtype=np.uint8)        # a sequence every ~8 bytes: 4-byte matches at random near offsets
    for p_ in range(64, B - 8, 8):
        o = int(rng.integers(4, 60))
        a[p_:p_ + 4] = a[p_ - o:p_ - o + 4]
    blocks.append(a)
    comps.append(oracle.lz4_compress(a, 1))
    with _indexed(codec, 2):
        _decode_check(codec, comps, blocks, B, ("two walkers", B))


@pytest.mark.parametrize("walkers", WALKERS)
def test_indexed_decoder_golden_cells(codec, oracle, walkers):
    """tests/golden/vectors.json, LZ4 cells at 128 KiB and 1 MiB: streams of liblz4 1.9.3 (hash-pinned: the oracle's
    bytes hash to comp_sha256) decoded by the indexed path hash to raw_sha256"""
    import hashlib
    import json
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    cells = [c for c in json.load(open(os.path.join(G, "vectors.json")))["cells"]
             if c["method"] == "lz4" and c["B"] in (131072, 1 << 20)]
    assert len(cells) >= 100
    with _indexed(codec, walkers):
        for B in (131072, 1 << 20):
            sub = [c for c in cells if c["B"] == B]
            if walkers != 1:
                sub = sub[::3]
            comps = []
            for c in sub:
                k = oracle.lz4_compress(oracle.synth(0, c["block"], B, c["dist"]), c["param"])
                assert len(k) == c["csize"] and sha(k) == c["comp_sha256"], c
                comps.append(k)
            outs, st = codec.decompress_blocks(METHOD_LZ4, comps, B)
            assert (st == 0).all()
            for c, o in zip(sub, outs):
                assert sha(o) == c["raw_sha256"], (c, walkers)


@pytest.mark.parametrize("walkers", WALKERS)
def test_indexed_decoder_structured_and_stock_streams(codec, oracle, walkers):
    """structured random blocks (tests/stress_gpu.py: repeats at any distance, runs, word mixtures, noise, short
    periods) of 128 KiB ... 1 MiB, compressed by the STOCK liblz4 where it can be loaded (else by the oracle, which
    is pinned to it): positions that wrap 16 bits, far matches against the 4 KiB ring, long runs, periodic data on
    which index walkers started at guessed positions never meet the true chain (the serial re-walk)"""
    import oracle_lib
    from stress_gpu import make_block
    stock = oracle_lib.StockLibs()
    comp = stock.lz4_compress if stock.lz4 is not None else oracle.lz4_compress
    rng = np.random.default_rng(1234 + walkers)
    for B in (131072, 400000, 1 << 20):
        blocks = [make_block(rng, B) for _ in range(6)]
        # strictly periodic text: every guessed start lands in the same phase of the period
        blocks.append(np.resize(np.frombuffer(b"0123456789abcdefghijklmnopqrstu;", np.uint8), B).copy())
        t = np.resize(np.frombuffer(b"\x10\x10\x10\x10\x10\x10\x10A", np.uint8), B).copy()
        t[::4099] = rng.integers(0, 256, len(t[::4099]), dtype=np.uint8)
        blocks.append(t)
        blocks.append(rng.integers(0, 256, B, dtype=np.uint8))                     # one literal run
        z = np.zeros(B, np.uint8); z[B // 2] = 7
        blocks.append(z)                                                             # two long matches
        comps = [comp(b, int(rng.integers(1, 9))) for b in blocks]
        with _indexed(codec, walkers):
            _decode_check(codec, comps, blocks, B, ("structured", B, walkers))


@pytest.mark.parametrize("B", [131072, 1 << 20])
@pytest.mark.parametrize("walkers", WALKERS)
def test_indexed_decoder_mutated_streams(codec, oracle, B, walkers):
    """corrupted 128 KiB / 1 MiB streams: verdict and bytes as the oracle's (success iff liblz4 decodes exactly B bytes)"""
    from stress_gpu import mutate
    rng = np.random.default_rng(77 + walkers)
    comps, expect = [], []
    base = [oracle.lz4_compress(oracle.synth(9, d, B, d), 1 + 6 * (d & 1)) for d in (0, 1, 2, 3)]
    n_mut = 60 if B == 131072 else 16
    for it in range(n_mut):
        m = mutate(rng, base[it % len(base)])
        r, out = oracle.lz4_decompress(m, B, fill=0xA5)
        comps.append(m)
        expect.append(out.copy() if r == B else None)
    comps += base
    expect += [oracle.synth(9, d, B, d) for d in (0, 1, 2, 3)]
    with _indexed(codec, walkers):
        _decode_check(codec, comps, expect, B, ("mutated", B, walkers))
    assert sum(e is None for e in expect) >= n_mut // 4


def test_indexed_decoder_packed_offsets_and_ragged_batch(codec, oracle):
    """streams packed at arbitrary byte offsets (the index stages whole 128-byte lines), a batch that is not a multiple
    of the walkers' wave width, empty streams among them; walkers 8 as the 1 MiB configuration of 8192 blocks gets"""
    from pg_cryogen_amd import codec as cc
    B = 131072
    blocks = [oracle.synth(23, i, B, i % 5) for i in range(11)]
    comps = [oracle.lz4_compress(b, 1) for b in blocks] + [np.zeros(0, np.uint8)]
    n = len(comps)
    offs, pos = [], 77
    for c in comps:
        offs.append(pos)
        pos += len(c) + 1 + (len(c) % 13)
    packed = np.full(pos + 64, 0xEE, np.uint8)
    for o, c in zip(offs, comps):
        packed[o:o + len(c)] = c
    d_src, d_off, d_sz = codec.alloc(packed.nbytes), codec.alloc(8 * n), codec.alloc(4 * n)
    d_dst, d_st = codec.alloc(n * B), codec.alloc(4 * n)
    d_src.upload(packed)
    d_off.upload(np.array(offs, np.uint64))
    d_sz.upload(np.array([len(c) for c in comps], np.uint32))
    for walkers in (1, 2, 8, 32):
        d_dst.memset(0x5A)
        with _indexed(codec, walkers):
            codec.decompress_batch(METHOD_LZ4, d_src, d_off, d_sz, d_dst, B, B, n, d_st)
            codec.sync()
        st = d_st.download(dtype=np.int32)
        raw = d_dst.download().reshape(n, B)
        for i, b in enumerate(blocks):
            assert st[i] == 0 and np.array_equal(raw[i], b), (walkers, i)
        assert st[n - 1] != 0
    for x in (d_src, d_off, d_sz, d_dst, d_st):
        x.free()


# ---------------------------------------------------------------------------------------------------------------
# The few-blocks path (lz4_lat.hip: every output byte in parallel; what one block per call -- the reference's own call
# shape, pg_cryogen.c:726, cache.c:178 -- and its 16 cache slots get).  Forced by option so that the batches below take it
# whatever the automatic choice becomes; anything it does not decode itself goes to the batch decoder in the same call,
# so verdicts and bytes must be the oracle's for corrupted streams too.
# ---------------------------------------------------------------------------------------------------------------
def _few_blocks(codec):
    from pg_cryogen_amd import codec as cc

    class _Ctx:
        def __enter__(self_):
            codec.set_option(cc.OPT_LZ4_DECODE_PATH, cc.LZ4_PATH_FEW_BLOCKS)

        def __exit__(self_, *a):
            codec.set_option(cc.OPT_LZ4_DECODE_PATH, cc.LZ4_PATH_AUTO)
    return _Ctx()


@pytest.mark.parametrize("B", [32768, 131072, 400000, 1 << 20])
def test_few_blocks_path_distributions(codec, oracle, B):
    """all five distributions x accelerations 1 / 7 / 50, one block per call and 15 per call"""
    blocks, comps = [], []
    for d in range(5):
        for a in (1, 7, 50):
            b = oracle.synth(31 + a, d, B, d)
            blocks.append(b)
            comps.append(oracle.lz4_compress(b, a))
    with _few_blocks(codec):
        _decode_check(codec, comps, blocks, B, ("few", B))
        for i in (0, 4, 9):
            _decode_check(codec, comps[i:i + 1], blocks[i:i + 1], B, ("one", B, i))


def test_few_blocks_path_golden_cells(codec, oracle):
    """tests/golden/vectors.json, LZ4 cells at 128 KiB and 1 MiB, 16 streams per call"""
    import hashlib
    import json
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    cells = [c for c in json.load(open(os.path.join(G, "vectors.json")))["cells"]
             if c["method"] == "lz4" and c["B"] in (131072, 1 << 20)]
    assert len(cells) >= 100
    with _few_blocks(codec):
        for B in (131072, 1 << 20):
            sub = [c for c in cells if c["B"] == B][::2]
            for k in range(0, len(sub), 16):
                part = sub[k:k + 16]
                comps = []
                for c in part:
                    m = oracle.lz4_compress(oracle.synth(0, c["block"], B, c["dist"]), c["param"])
                    assert len(m) == c["csize"] and sha(m) == c["comp_sha256"], c
                    comps.append(m)
                outs, st = codec.decompress_blocks(METHOD_LZ4, comps, B)
                assert (st == 0).all()
                for c, o in zip(part, outs):
                    assert sha(o) == c["raw_sha256"], c


@pytest.mark.parametrize("B", [131072, 1 << 20])
def test_few_blocks_path_structured_and_mutated(codec, oracle, B):
    """structured blocks (repeats at any distance, long runs, periodic data), stock-library streams where liblz4 is
    there, and corrupted streams: verdict and bytes as the oracle's"""
    from stress_gpu import make_block, mutate
    rng = np.random.default_rng(4242 + B)
    blocks = [make_block(rng, B) for _ in range(10)]
    blocks.append(np.tile(np.arange(7, dtype=np.uint8), B // 7 + 1)[:B].copy())      # strictly periodic
    blocks.append(np.zeros(B, np.uint8))
    blocks.append(rng.integers(0, 256, B, dtype=np.uint8))                            # incompressible: left to the parser
    comps = [oracle.lz4_compress(b, int(rng.integers(1, 40))) for b in blocks]
    expect = list(blocks)
    for it in range(24 if B == 131072 else 10):
        m = mutate(rng, comps[it % 10])
        r, out = oracle.lz4_decompress(m, B, fill=0xA5)
        comps.append(m)
        expect.append(out.copy() if r == B else None)
    with _few_blocks(codec):
        for k in range(0, len(comps), 16):
            _decode_check(codec, comps[k:k + 16], expect[k:k + 16], B, ("few-mutated", B, k))
    assert sum(e is None for e in expect) >= 4


def test_few_blocks_path_edge_streams(codec, oracle):
    """what the few-blocks path must leave to the batch decoder, or decode at the edges of its rules: an empty stream, a
    stream of literals only, liblz4's offset-0 zero fill, a match that ends exactly LASTLITERALS before the end, one that
    ends one byte later (rejected), a truncated stream, trailing garbage; one block per call and all in one call"""
    B = 32768

    def lits_len(n):   # token nibble 15 + extension bytes
        out, n = [], n - 15
        while n >= 255:
            out.append(255)
            n -= 255
        out.append(n)
        return out

    def stream(ll1, off, ml, tail):
        """literals ll1 x 'a' | match (off, ml) | literals `tail`"""
        b = bytearray()
        b.append(((15 if ll1 >= 15 else ll1) << 4) | (15 if ml - 4 >= 15 else ml - 4))
        if ll1 >= 15:
            b += bytes(lits_len(ll1))
        b += b"a" * ll1
        b += bytes([off & 255, off >> 8])
        if ml - 4 >= 15:
            b += bytes(lits_len(ml - 4))
        b.append((15 if tail >= 15 else tail) << 4)
        if tail >= 15:
            b += bytes(lits_len(tail))
        b += bytes((i * 7 + 1) & 255 for i in range(tail))
        return np.frombuffer(bytes(b), np.uint8).copy()

    rng = np.random.default_rng(99)
    items = [np.zeros(0, np.uint8),                                                      # empty
             oracle.lz4_compress(rng.integers(0, 256, B, dtype=np.uint8), 1),           # literals only
             stream(1, 0, B - 1 - 12, 12),                                               # offset 0: zero fill
             stream(20, 3, B - 20 - 5, 5),                                               # match ends LASTLITERALS before the end
             stream(20, 3, B - 20 - 4, 4),                                               # ... one byte later: rejected
             stream(20, 21, B - 20 - 12, 12),                                            # offset beyond the output so far: rejected
             oracle.lz4_compress(oracle.synth(3, 1, B, 0), 1)[:-7],                      # truncated
             np.concatenate([oracle.lz4_compress(oracle.synth(3, 2, B, 0), 1), np.array([0, 0, 0], np.uint8)])]  # trailing bytes
    expect = []
    for m in items:
        r, out = oracle.lz4_decompress(m, B, fill=0xA5)
        expect.append(out.copy() if r == B else None)
    assert expect[2] is not None and expect[3] is not None and expect[4] is None and expect[5] is None and expect[0] is None
    with _few_blocks(codec):
        _decode_check(codec, items, expect, B, "few-edges")
        for i in range(len(items)):
            _decode_check(codec, items[i:i + 1], expect[i:i + 1], B, ("few-edge-one", i))


def test_few_blocks_path_many_walkers_corners(codec, oracle):
    """Round 5: the few-blocks index runs up to 1 024 direct-read walkers per block in two kernels (k_lz4_few_walk /
    k_lz4_few_join).  Corners of ITS rules at 1 MiB and 2 MiB blocks: compressed sizes around the powers of two that set the
    number of walkers, a block whose sequences are all three bytes long (more tokens per segment than a row holds: the block
    must fall back whole), segments that lie inside one long literal run or one long 255-run (chains that cannot meet), and
    a stream cut in the middle of a segment.  Verdicts and bytes as the oracle's, one block per call and together."""
    rng = np.random.default_rng(20261004)
    for B in (1 << 20, 2 << 20):
        blocks = []
        # 3-byte sequences: a 4-byte pattern repeated with one byte changed every 7 bytes -> tokens every 3-4 bytes of stream
        a = np.tile(np.array([1, 2, 3, 4, 5, 6, 7], np.uint8), B // 7 + 1)[:B].copy()
        a[::11] ^= rng.integers(1, 255, len(a[::11]), dtype=np.uint8)
        blocks.append(a)
        # a long literal run in the middle of compressible data, and a long run of one byte (a 255-run of match length)
        b = oracle.synth(77, 1, B, 0).copy()
        b[B // 3:B // 3 + 40000] = rng.integers(0, 256, 40000, dtype=np.uint8)
        blocks.append(b)
        c = oracle.synth(77, 2, B, 0).copy()
        c[B // 2:B // 2 + 300000] = 0x5A
        blocks.append(c)
        blocks.append(oracle.synth(77, 3, B, 1))                      # narrow rows: 84 % zero gap
        comps = [oracle.lz4_compress(x, 1) for x in blocks]
        expect = list(blocks)
        # compressed sizes right at the walker-count steps: cut a valid stream's INPUT so that it compresses to about 2^k KiB
        wide78 = oracle.synth(78, 0, B, 0)
        for target in (1 << 19, (1 << 19) + 1024):
            lo, hi = 1000, B
            while hi - lo > 64:                                        # bisection on the number of `wide` bytes kept
                mid = (lo + hi) // 2
                x = np.zeros(B, np.uint8)
                x[:mid] = wide78[:mid]
                if len(oracle.lz4_compress(x, 1)) < target:
                    lo = mid
                else:
                    hi = mid
            x = np.zeros(B, np.uint8)
            x[:lo] = wide78[:lo]
            comps.append(oracle.lz4_compress(x, 1))
            expect.append(x)
        cut = comps[1][:len(comps[1]) // 2 + 123].copy()               # truncated inside a segment
        r, out = oracle.lz4_decompress(cut, B, fill=0xA5)
        comps.append(cut)
        expect.append(out.copy() if r == B else None)
        assert expect[-1] is None
        with _few_blocks(codec):
            _decode_check(codec, comps, expect, B, ("few-corners", B))
            for i in range(len(comps)):
                _decode_check(codec, comps[i:i + 1], expect[i:i + 1], B, ("few-corner-one", B, i))


def test_indexed_decoder_last_round_on_the_side_stream(codec, oracle):
    """Round 5: a batch of between one and two rounds of the one-wave decoder (6 144 blocks are resident) sends its last,
    partial round to a low-priority side stream with two waves per block.  7 000 blocks of 32 KiB made and compressed on the
    device; streams corrupted in the first part, at the seam and in the last part; every block compared with its original on
    the device, the corrupted ones' verdicts with the oracle's, 40 sampled blocks byte for byte with the oracle."""
    from pg_cryogen_amd import bound
    B, n = 32768, 7000
    stride = (bound(METHOD_LZ4, B) + 15) & ~15
    d_raw, d_comp, d_out = codec.alloc(n * B), codec.alloc(n * stride), codec.alloc(n * B)
    d_sizes, d_off, d_st, d_mis = codec.alloc(4 * n), codec.alloc(8 * n), codec.alloc(4 * n), codec.alloc(8)
    try:
        codec.synth_batch(13, 0, n, B, 0, d_raw)
        codec.compress_batch(METHOD_LZ4, 1, d_raw, B, B, n, d_comp, stride, d_sizes, d_st)
        assert (d_st.download(dtype=np.int32) == 0).all()
        sizes = d_sizes.download(dtype=np.uint32)
        d_off.upload(np.arange(n, dtype=np.uint64) * np.uint64(stride))
        bad = [5, 3071, 6143, 6144, 6145, 6999]
        for i in bad:                                                   # an offset beyond the output so far, early in the stream
            c = d_comp.download(int(sizes[i]), offset=i * stride)
            c[40:48] = 0xFF
            r, _ = oracle.lz4_decompress(c, B)
            assert r != B
            d_comp.upload(c, offset=i * stride)
        d_out.memset(0xEE)
        codec.decompress_batch(METHOD_LZ4, d_comp, d_off, d_sizes, d_out, B, B, n, d_st)
        codec.sync()
        st = d_st.download(dtype=np.int32)
        assert all(st[i] != 0 for i in bad) and int((st != 0).sum()) == len(bad), np.nonzero(st)[0][:20]
        for i in sorted(set(list(range(0, n, n // 32)) + [6142, 6146, 6998])):
            if i in bad:
                continue
            raw = oracle.synth(13, i, B, 0)
            comp = d_comp.download(int(sizes[i]), offset=i * stride)
            r, exp = oracle.lz4_decompress(comp, B)
            assert r == B and np.array_equal(exp, raw) and np.array_equal(d_out.download(B, offset=i * B), exp), i
        # everything else against the originals, on the device: the corrupted blocks are the only mismatches
        d_mis.memset(0)
        codec.compare_batch(d_raw, B, d_out, B, B, n, d_mis)
        codec.sync()
        assert 0 < int(d_mis.download(dtype=np.uint64)[0]) <= len(bad) * B
    finally:
        for x in (d_raw, d_comp, d_out, d_sizes, d_off, d_st, d_mis):
            x.free()


def test_lz4_path_options_roundtrip(codec):
    from pg_cryogen_amd import codec as cc
    assert codec.get_option(cc.OPT_LZ4_DECODE_PATH) == 0 and codec.get_option(cc.OPT_LZ4_INDEX_WALKERS) == 0
    codec.set_option(cc.OPT_LZ4_INDEX_WALKERS, 16)
    assert codec.get_option(cc.OPT_LZ4_INDEX_WALKERS) == 16
    codec.set_option(cc.OPT_LZ4_INDEX_WALKERS, 0)
    with pytest.raises(cc.CryoError):
        codec.set_option(cc.OPT_LZ4_INDEX_WALKERS, 3)        # not a power of two
    with pytest.raises(cc.CryoError):
        codec.set_option(cc.OPT_LZ4_DECODE_PATH, 9)
