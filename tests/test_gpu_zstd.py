"""GPU parity tests for the zstd decode path: HIP kernel through the C ABI vs the CPU oracle
and the golden streams produced by libzstd 1.4.8.

Reference call shape: compression.c:116 ZSTD_decompress(out, B, src, csize).  Bar: bit-exact.
"""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib
from pg_cryogen_amd import METHOD_ZSTD

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(params=["pipeline", "fused"])
def zstd_path(request, codec):
    """both decoders, whatever the batch size: the four-kernel pipeline (what every batch takes by default) and the fused
    one-wave-per-frame kernel (what the pipeline hands its irregular frames to)"""
    from pg_cryogen_amd import codec as cc
    codec.set_option(cc.OPT_ZSTD_DECODE_PATH, 2 if request.param == "pipeline" else 1)
    yield request.param
    codec.set_option(cc.OPT_ZSTD_DECODE_PATH, 0)


def test_zstd_decode_golden_streams(codec, zstd_path):
    streams = json.load(open(os.path.join(G, "streams.json")))["streams"]
    by_B = {}
    for s in streams:
        if s["method"] == "zstd":
            by_B.setdefault(s["B"], []).append(s)
    assert by_B
    for B, lst in by_B.items():
        comps = [np.frombuffer(base64.b64decode(s["data"]), np.uint8) for s in lst]
        outs, st = codec.decompress_blocks(METHOD_ZSTD, comps, B)
        assert (st == 0).all(), (B, st)
        for s, o in zip(lst, outs):
            assert sha(o) == s["raw_sha256"], (B, s["dist"], s["param"])


def test_zstd_decode_adversarial_matches_oracle(codec, oracle, zstd_path):
    """malformed streams: the kernel's verdict and bytes equal the oracle's (which is pinned to
    libzstd by tests/test_oracle_golden.py)"""
    adv = json.load(open(os.path.join(G, "adversarial.json")))["cases"]
    cases = [c for c in adv if c["method"] == "zstd" and len(c["data"]) > 0]
    B = cases[0]["B"]
    comps = [np.frombuffer(base64.b64decode(c["data"]), np.uint8) for c in cases]
    outs, st = codec.decompress_blocks(METHOD_ZSTD, comps, B)
    n_ok = 0
    for c, m, o, s in zip(cases, comps, outs, st):
        r, exp = oracle.zstd_decompress(m, B, fill=0xA5)
        if r == B:
            assert s == 0, c["name"]
            assert np.array_equal(o, exp), c["name"]
            n_ok += 1
        else:
            assert s != 0, c["name"]
    assert n_ok >= 3


@pytest.mark.parametrize("B", [131072, 1 << 20, 4096, 65546, 300])
def test_zstd_decode_all_levels_live_library(codec, oracle, B, zstd_path):
    stock = oracle_lib.StockLibs()
    if stock.zstd is None:
        pytest.skip("libzstd.so.1 not loadable")
    blocks, comps = [], []
    for dist in range(5):
        raw = oracle.synth(9, dist, B, dist)
        for lvl in (-5, 1, 2, 3, 5, 9, 16, 19, 22):
            if B >= 131072 and lvl > 9 and dist in (0, 3):
                continue
            if B == (1 << 20) and lvl > 5:
                continue
            blocks.append(raw)
            comps.append(stock.zstd_compress(raw, lvl))
    outs, st = codec.decompress_blocks(METHOD_ZSTD, comps, B)
    assert (st == 0).all(), st
    for raw, o in zip(blocks, outs):
        assert np.array_equal(o, raw)


def test_zstd_decode_literal_sections_around_the_walker_threshold(codec, oracle, zstd_path):
    """Huffman blocks of fewer than 4 KiB of literals are decoded lane per stream (k_zhuf), longer ones by k_zhufw's walkers
    (zstd_pipe.hip, `hufw_min`): literal sections of 3 900 ... 4 300 bytes in one batch -- random hex that matches nothing, then a
    zero gap --, short sections too (40 ... 1 100 bytes: raw, one Huffman stream, four); and
    mutated copies whose verdicts and bytes must be the oracle's on either side of the threshold."""
    rng = np.random.default_rng(4096)
    hexd = np.frombuffer(b"0123456789abcdef", np.uint8)
    B = 16384
    blocks = []
    for n in list(range(3900, 4300, 12)) + list(range(40, 260, 30)) + list(range(200, 1100, 60)) + [4095, 4096, 4097]:
        b = np.zeros(B, np.uint8)
        b[:n] = hexd[rng.integers(0, 16, n)]
        blocks.append(b)
    comps = [oracle.zstd_compress(b, 1) for b in blocks]
    outs, st = codec.decompress_blocks(METHOD_ZSTD, comps, B)
    assert (st == 0).all(), st
    for i, (raw, o) in enumerate(zip(blocks, outs)):
        assert np.array_equal(o, raw), i
    from stress_gpu import mutate
    items, expect = [], []
    for it in range(60):
        m = mutate(rng, comps[it % len(comps)])
        r, out = oracle.zstd_decompress(m, B, fill=0xA5)
        items.append(m)
        expect.append(out.copy() if r == B else None)
    outs, st = codec.decompress_blocks(METHOD_ZSTD, items, B)
    for i, e in enumerate(expect):
        if e is None:
            assert st[i] != 0, i
        else:
            assert st[i] == 0 and np.array_equal(outs[i], e), i


@pytest.mark.parametrize("B", [131072, 1 << 20])
def test_zstd_decode_literals_of_one_code_length(codec, oracle, B, zstd_path):
    """Literal streams whose Huffman codes (nearly) all have one length -- hex digits, decimal digits, base64 -- are what
    the Huffman walkers of k_zhufw synchronise slowest on (chains at different offsets modulo the code length stay apart
    until a rarer, longer code comes by): 1 MiB frames of hex-heavy rows had 61 % of their blocks handed back to k_zhuf by
    the first version.  Random symbols of such alphabets with a sprinkling of others, so that nothing matches and
    (almost) everything is a literal; decoded bytes must be the input whichever kernel ends up decoding a block."""
    rng = np.random.default_rng(B + 5)
    alphabets = [b"0123456789abcdef", b"0123456789", b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/",
                 b"01", b"0123456789abcdef-{}\":, "]
    blocks = []
    for k, al in enumerate(alphabets):
        a = np.frombuffer(al, np.uint8)
        for rare in (0.0, 0.002, 0.03):
            b = a[rng.integers(0, len(a), B)]
            m = rng.random(B) < rare
            b = np.where(m, rng.integers(0, 256, B).astype(np.uint8), b).astype(np.uint8)
            blocks.append(np.ascontiguousarray(b))
    comps = [oracle.zstd_compress(b, 1 if i % 2 == 0 else 3) for i, b in enumerate(blocks)]
    outs, st = codec.decompress_blocks(METHOD_ZSTD, comps, B)
    assert (st == 0).all(), st
    for i, (raw, o) in enumerate(zip(blocks, outs)):
        assert np.array_equal(o, raw), i
    # the same streams corrupted: verdict and bytes as the oracle's (the walkers hand a block back to the lane-per-stream
    # kernel whenever their counts do not add up, so a verdict is that kernel's)
    from stress_gpu import mutate
    items, expect = [], []
    for it in range(40 if B == 131072 else 12):
        m = mutate(rng, comps[it % len(comps)])
        r, out = oracle.zstd_decompress(m, B, fill=0xA5)
        items.append(m)
        expect.append(out.copy() if r == B else None)
    outs, st = codec.decompress_blocks(METHOD_ZSTD, items, B)
    for i, e in enumerate(expect):
        if e is None:
            assert st[i] != 0, i
        else:
            assert st[i] == 0 and np.array_equal(outs[i], e), i


def test_zstd_decode_fuzz_matches_oracle(codec, oracle, zstd_path):
    stock = oracle_lib.StockLibs()
    if stock.zstd is None:
        pytest.skip("libzstd.so.1 not loadable")
    rng = np.random.default_rng(21)
    B = 4096
    comps, expect = [], []
    for dist in (0, 1, 3):
        for lvl in (1, 3, 19):
            c = stock.zstd_compress(oracle.synth(0, 1, B, dist), lvl)
            for it in range(200):
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
                    p = int(rng.integers(0, len(m)))
                    m[p] ^= 1 << int(rng.integers(0, 8))
                r, out = oracle.zstd_decompress(m, B, fill=0xA5)
                comps.append(m)
                expect.append(out.copy() if r == B else None)
    outs, st = codec.decompress_blocks(METHOD_ZSTD, comps, B)
    n_ok = 0
    for i, e in enumerate(expect):
        if e is None:
            assert st[i] != 0, i
        else:
            assert st[i] == 0, i
            assert np.array_equal(outs[i], e), i
            n_ok += 1
    assert n_ok > 30


def _drop_content_size(c):
    """the same frame without Frame_Content_Size: not single-segment, a window descriptor instead (what a streaming
    compressor that was not told the size writes); the blocks are untouched"""
    fhd = int(c[4])
    fcs_flag, single, did = fhd >> 6, (fhd >> 5) & 1, fhd & 3
    fcs_bytes = [1 if single else 0, 2, 4, 8][fcs_flag]
    hdr = 5 + (0 if single else 1) + [0, 1, 2, 4][did] + fcs_bytes
    wd = (21 - 10) << 3                                        # 2 MiB window: covers every block size used here
    return np.concatenate([c[:4], np.array([fhd & 0x04, wd], np.uint8), c[hdr:]])   # keep only the checksum flag (0 here)


@pytest.mark.parametrize("B", [131072, 1 << 20])
def test_zstd_decode_few_frames_path_matches_oracle(codec, oracle, B):
    """the byte-parallel execution that calls of up to 64 frames take by default (k_zlat_* + lat_copy.h) against
    sequence-rich frames -- `wide` rows at levels 1 / 3 / 9, with and without content size, frames with raw and RLE blocks,
    mutated copies -- at 1 .. 64 frames per call: verdicts and bytes == oracle, with the path on (automatic) and off
    (CRYO_OPT_ZSTD_DECODE_PATH = 3: k_zexec for every frame)."""
    from pg_cryogen_amd import codec as cc
    stock = oracle_lib.StockLibs()
    if stock.zstd is None:
        pytest.skip("libzstd.so.1 not loadable")
    rng = np.random.default_rng(77)
    raws = [oracle.synth(9, i, B, 0) for i in range(4)]
    if B > 131072:   # a frame of several zstd blocks, some of them raw (noise) and RLE (a run of one byte)
        mix = oracle.synth(9, 9, B, 0).copy()
        mix[131072:262144] = rng.integers(0, 256, 131072, dtype=np.uint8)
        mix[393216:655360] = 0x5A
        raws.append(mix)
    frames = []
    for lvl in (1, 3, 9):
        for r in raws:
            c = stock.zstd_compress(r, lvl)
            frames += [c, _drop_content_size(c)]
    assert any(int(f[4]) >> 6 == 0 and not (int(f[4]) >> 5) & 1 for f in frames)
    muts = []
    for c in frames[::3]:
        for _ in range(6):
            m = c.copy()
            k = int(rng.integers(0, 3))
            if k == 0:
                m[int(rng.integers(8, len(m)))] ^= 1 << int(rng.integers(0, 8))
            elif k == 1:
                m = m[:int(rng.integers(len(m) // 2, len(m)))].copy()
            else:
                p = int(rng.integers(8, len(m) - 4))
                m[p:p + 3] = rng.integers(0, 256, 3, dtype=np.uint8)
            muts.append(m)
    comps = frames + muts
    expect = []
    for m in comps:
        r, out = oracle.zstd_decompress(m, B, fill=0xA5)
        expect.append(out.copy() if r == B else None)
    assert sum(e is not None for e in expect) >= len(frames)      # every unmutated frame decodes
    sizes = (1, 3, 16, 64) if B == 131072 else (1, 5, 16)
    try:
        for path in (0, 3):
            codec.set_option(cc.OPT_ZSTD_DECODE_PATH, path)
            for k in sizes:
                for first in range(0, len(comps), k):
                    part, exp = comps[first:first + k], expect[first:first + k]
                    outs, st = codec.decompress_blocks(METHOD_ZSTD, part, B)
                    for i, e in enumerate(exp):
                        if e is None:
                            assert st[i] != 0, (path, k, first + i)
                        else:
                            assert st[i] == 0 and np.array_equal(outs[i], e), (path, k, first + i)
    finally:
        codec.set_option(cc.OPT_ZSTD_DECODE_PATH, 0)


def test_zstd_single_block_host_api(codec, oracle):
    stock = oracle_lib.StockLibs()
    if stock.zstd is None:
        pytest.skip("libzstd.so.1 not loadable")
    B = 131072
    raw = oracle.synth(2, 9, B, 0)
    c = stock.zstd_compress(raw, 1)
    out = codec.decompress_block(METHOD_ZSTD, c, B)
    assert np.array_equal(out, raw)
    assert codec.decompress_block(METHOD_ZSTD, c[:-3], B) is None


# ---------------- zstd encode (every strategy, `fast` .. `btultra2`: levels -5..22), reference compression.c:102-104 ----------------
@pytest.mark.parametrize("B", [131072, 1 << 20, 65546, 20000])
def test_zstd_encode_bit_exact(codec, oracle, B):
    blocks = [oracle.synth(3, blk, B, dist) for dist in range(5) for blk in range(2)]
    for lvl in (-5, -1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10):   # level 6 is `greedy` above 256 KiB, `lazy` below; 7+ `lazy` / `lazy2`
        got = codec.compress_blocks(METHOD_ZSTD, lvl, blocks)
        for i, b in enumerate(blocks):
            exp = oracle.zstd_compress(b, lvl)
            assert len(exp) > 0
            assert len(got[i]) == len(exp), (B, lvl, i, len(got[i]), len(exp))
            assert np.array_equal(got[i], exp), (B, lvl, i)


@pytest.mark.parametrize("B", [64, 1000, 4096, 16384, 16385, 131073, 200000, 262144, 262145, 1 << 20])
def test_zstd_encode_every_size_class_bit_exact(codec, oracle, B):
    """libzstd picks its parameters from four tables by source size (<= 16 KiB, <= 128 KiB, <= 256 KiB, above): every
    class, every level below the optimal-parser strategies there (-5 .. 10 / 12 / 12 / 15) and btopt / btultra / btultra2
    above them (all of them up to 16 KiB, a sample on the larger sizes), against the oracle (pinned to libzstd 1.4.8 on these
    sizes by tests/test_oracle_golden.py) and, where loadable, the stock library"""
    from stress_gpu import make_block
    from test_oracle_golden import zstd_levels_with_kernel
    from pg_cryogen_amd.codec import CryoError, E_UNSUPPORTED
    stock = oracle_lib.StockLibs()
    rng = np.random.default_rng(B + 1)
    blocks = [make_block(rng, B), oracle.synth(6, 5, B, 0) if B >= 4096 else rng.integers(0, 4, B, dtype=np.uint8),
              rng.integers(0, 256, B, dtype=np.uint8)]
    levels = zstd_levels_with_kernel(B)
    for lvl in levels:
        got = codec.compress_blocks(METHOD_ZSTD, lvl, blocks)
        for i, b in enumerate(blocks):
            exp = oracle.zstd_compress(b, lvl)
            assert len(exp) > 0 and np.array_equal(got[i], exp), (B, lvl, i, len(got[i]), len(exp))
            if stock.zstd is not None and (lvl in (levels[0], 1) or lvl >= 11):
                assert np.array_equal(got[i], stock.zstd_compress(b, lvl)), (B, lvl, i)
    outs, st = codec.decompress_blocks(METHOD_ZSTD, got, B)
    assert (st == 0).all() and all(np.array_equal(o, b) for o, b in zip(outs, blocks))
    with pytest.raises(CryoError) as e:
        codec.compress_blocks(METHOD_ZSTD, 23, blocks[:1])   # above ZSTD_maxCLevel
    assert e.value.code == E_UNSUPPORTED


def test_zstd_encode_blocks_of_few_sequences(codec, oracle):
    """the soak's block of four sequences (tests/golden/soak_block_four_sequences.npy) and periodic blocks like it: sequence
    tables over very few symbols, every strategy from `greedy` up"""
    from test_oracle_golden import few_sequence_blocks
    b = np.load(os.path.join(G, "soak_block_four_sequences.npy"))
    for lvl in (5, 7, 10, 13, 22):
        got = codec.compress_blocks(METHOD_ZSTD, lvl, [b])
        assert np.array_equal(got[0], oracle.zstd_compress(b, lvl)), lvl
    blocks = few_sequence_blocks(321, 60)
    by_size = {}
    for blk in blocks:
        by_size.setdefault(len(blk), []).append(blk)
    for lvl in (5, 9, 12, 13, 17, 22):
        for n, group in by_size.items():
            got = codec.compress_blocks(METHOD_ZSTD, lvl, group)
            for g, blk in zip(got, group):
                assert np.array_equal(g, oracle.zstd_compress(blk, lvl)), (n, lvl)


def test_zstd_encode_matches_golden_vectors(codec, oracle):
    """every golden zstd cell (levels -5 .. 7 and 22), every size; of the level-22 cells at 1 MiB (seconds per block) one block
    per distribution"""
    cells = [c for c in json.load(open(os.path.join(G, "vectors.json")))["cells"]
             if c["method"] == "zstd" and not (c["param"] == 22 and c["B"] > 131072 and c["block"] > 0)]
    assert len(cells) >= 360 and sum(c["param"] == 22 for c in cells) >= 30
    for B, lvl in sorted(set((c["B"], c["param"]) for c in cells)):
        sub = [c for c in cells if c["param"] == lvl and c["B"] == B]
        blocks = [oracle.synth(0, c["block"], c["B"], c["dist"]) for c in sub]
        got = codec.compress_blocks(METHOD_ZSTD, lvl, blocks)
        for c, g in zip(sub, got):
            assert len(g) == c["csize"] and sha(g) == c["comp_sha256"], c


def test_zstd_roundtrip_on_device_and_unsupported_levels(codec, oracle):
    from pg_cryogen_amd.codec import CryoError, E_UNSUPPORTED
    B = 131072
    blocks = [oracle.synth(8, i, B, i % 5) for i in range(10)]
    comps = codec.compress_blocks(METHOD_ZSTD, 1, blocks)
    outs, st = codec.decompress_blocks(METHOD_ZSTD, comps, B)
    assert (st == 0).all()
    for b, o in zip(blocks, outs):
        assert np.array_equal(b, o)
    with pytest.raises(CryoError) as e:
        codec.compress_blocks(METHOD_ZSTD, 23, blocks[:1])     # above ZSTD_maxCLevel (no CPU fallback)
    assert e.value.code == E_UNSUPPORTED


def test_zstd_decode_huffman_log12_crafted(codec, oracle):
    """2^12-entry Huffman table: fused decoder (small batch) and batch pipeline (two-level lookup)"""
    import zstd_craft
    for streams in (1, 4):
        frame, lits = zstd_craft.huf12_frame(n=700, streams=streams, seed=10 + streams)
        r, exp = oracle.zstd_decompress(frame, len(lits))
        assert r == len(lits) and np.array_equal(exp, lits)
        for n in (1, 40):   # 40 >= the pipeline threshold
            outs, st = codec.decompress_blocks(METHOD_ZSTD, [frame] * n, len(lits))
            assert (st == 0).all(), (streams, n, st)
            for o in outs:
                assert np.array_equal(o, lits), (streams, n)
        bad = frame.copy()
        bad[len(bad) // 2] ^= 0x10
        rb, _ = oracle.zstd_decompress(bad, len(lits))
        _, stb = codec.decompress_blocks(METHOD_ZSTD, [bad] * 40, len(lits))
        assert ((stb == 0).all() and rb == len(lits)) or ((stb != 0).all() and rb != len(lits))


def test_zstd_pipeline_many_frames_across_tiles(codec, oracle):
    """> 7680 frames: the batch pipeline runs two tiles; irregular inputs (concatenated frames, skippable
    frames, garbage, truncation) are sprinkled in and must get the fused decoder's verdicts"""
    B, n = 4096, 9000
    rng = np.random.default_rng(5)
    raws = [oracle.synth(3, i, B, i % 5) for i in range(40)]
    stock = oracle_lib.StockLibs()
    if stock.zstd is None:
        pytest.skip("needs the stock libzstd to produce 4 KiB frames (the oracle encoder mirrors the device's sizes)")
    comps = [stock.zstd_compress(r, 1 + (i % 5)) for i, r in enumerate(raws)]
    half = [stock.zstd_compress(r[:B // 2], 1) for r in raws[:4]]
    skippable = np.frombuffer((0x184D2A50).to_bytes(4, "little") + (5).to_bytes(4, "little") + b"hello", np.uint8)
    items, expect = [], []
    for i in range(n):
        k = int(rng.integers(0, 40))
        sel = i % 97
        if sel == 13:      # two concatenated frames decoding to B bytes
            a, b = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            items.append(np.concatenate([half[a], half[b]]))
            expect.append(np.concatenate([raws[a][:B // 2], raws[b][:B // 2]]))
        elif sel == 29:    # skippable frame in front
            items.append(np.concatenate([skippable, comps[k]]))
            expect.append(raws[k])
        elif sel == 41:    # truncated
            items.append(comps[k][:len(comps[k]) - 3].copy())
            expect.append(None)
        elif sel == 59:    # garbage
            items.append(rng.integers(0, 256, 100, dtype=np.uint8))
            expect.append(None)
        else:
            items.append(comps[k])
            expect.append(raws[k])
    outs, st = codec.decompress_blocks(METHOD_ZSTD, items, B)
    for i in range(n):
        if expect[i] is None:
            assert st[i] != 0, i
        else:
            assert st[i] == 0, i
            assert np.array_equal(outs[i], expect[i]), i


def test_zstd_encode_batch_match_finder_corners(codec, oracle):
    """the 64-iterations-per-step match finder: incompressible data (growing steps), repeats at distances
    beyond the LDS ring and the window, long runs (repeat-offset loop), periodic data with noise;
    block sizes at the edges of the supported ranges; every fast and dfast level.  Bar: bytes == oracle (== libzstd)."""
    rng = np.random.default_rng(21)
    stock = oracle_lib.StockLibs()
    for n in (16385, 20000, 131072, 262145, 300000 + 7):
        blocks = [rng.integers(0, 256, n, dtype=np.uint8)]
        a = rng.integers(0, 256, n, dtype=np.uint8)
        chunk = a[:3000].copy()
        for off in (4000, 4000 + 2047, 12000, n // 2, n - 3100):
            if off + 3000 <= n:
                a[off:off + 3000] = chunk
        blocks.append(a)
        z = np.zeros(n, np.uint8)
        z[n // 3:n // 3 + 100] = rng.integers(0, 256, 100, dtype=np.uint8)
        blocks.append(z)
        t = np.frombuffer((b"abcdefghij" * (n // 10 + 1))[:n], np.uint8).copy()
        t[::997] = rng.integers(0, 256, len(t[::997]), dtype=np.uint8)
        blocks.append(t)
        w = np.frombuffer((b"the quick brown fox jumps over the lazy dog, " * (n // 45 + 1))[:n], np.uint8).copy()
        w[rng.integers(0, n, n // 50)] = 0x5A
        blocks.append(w)
        for level in (-5, -1, 1, 2, 3, 4, 5, 6, 7, 8, 10):
            got = codec.compress_blocks(METHOD_ZSTD, level, blocks)
            for i, b in enumerate(blocks):
                exp = oracle.zstd_compress(b, level)
                assert len(exp) > 0
                assert np.array_equal(got[i], exp), (n, level, i, len(got[i]), len(exp))
                if stock.zstd is not None and level > 0:
                    assert np.array_equal(stock.zstd_compress(b, level), exp), (n, level, i)


def test_zstd_pipeline_dense_sequences_pool_overflow(codec, oracle):
    """streams with far more sequences than level 1 writes: text-like rows at levels 4/5 (B/8 sequences per frame)
    fit the batch pipeline's sequence pool (B/6 per frame); 4-byte words with a random byte between them at
    level 5 (B/5 per frame) overflow it, and the frames that do not fit must come out right through the
    irregular list (fused decoder)"""
    stock = oracle_lib.StockLibs()
    if stock.zstd is None:
        pytest.skip("needs the stock libzstd (levels above 4)")
    B = 131072
    rng = np.random.default_rng(5)
    words = rng.integers(0, 256, (64, 4), dtype=np.uint8)
    dense = []
    for k in range(6):
        a = np.zeros(B, np.uint8)
        idx = rng.integers(0, 64, B // 5)
        body = np.concatenate([words[idx], rng.integers(0, 256, (B // 5, 1), dtype=np.uint8)], axis=1).reshape(-1)
        a[:len(body)] = body
        dense.append(a)
    raws = [dense[i % 6] for i in range(40)] + [oracle.synth(9, i, B, 0) for i in range(8)]
    comps = [stock.zstd_compress(r, 5) for r in raws[:40]] + [stock.zstd_compress(r, 4 + (i & 1)) for i, r in enumerate(raws[40:])]
    outs, st = codec.decompress_blocks(METHOD_ZSTD, comps, B)
    assert (st == 0).all()
    for o, r in zip(outs, raws):
        assert np.array_equal(o, r)
