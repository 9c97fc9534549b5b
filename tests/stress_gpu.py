"""Differential stress run on the GPU box (not collected by pytest: `python tests/stress_gpu.py [seconds] [seed]`,
`python tests/stress_gpu.py fuzz [seconds] [seed]` for malformed streams).

Random structured blocks (mixtures of text-like rows, runs, repeats at random distances, noise) at several
block sizes, and every third round tiny / boundary-size / few-sequence / sparse blocks (odd_block); the device encoders must equal the stock liblz4 / libzstd byte for byte (all LZ4 accelerations,
zstd levels -5..22, the optimal-parser ones on a few blocks per round), and the device decoders must reproduce the input from streams the stock libraries wrote at
ANY level (zstd 1..19), through both zstd decode paths (fused for small batches, pipeline for large ones)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_lib
from pg_cryogen_amd import Codec, METHOD_LZ4, METHOD_ZSTD
from pg_cryogen_amd import codec as cc


def odd_block(rng, n):
    """what the structured generator does not make: periodic blocks with a few disturbed bytes (one to five sequences), sparse
    alphabets, stretched runs, uniform noise over a small alphabet"""
    kind = int(rng.integers(0, 4))
    if kind == 0:
        per = int(rng.choice([1, 2, 3, 4, 8, 13, 64]))
        blk = np.tile(rng.integers(0, 256, per, dtype=np.uint8), (n + per - 1) // per)[:n].copy()
        for _ in range(int(rng.integers(0, 5))):
            blk[int(rng.integers(0, n))] ^= int(rng.integers(1, 256))
        return blk
    if kind == 1:
        a = rng.integers(0, int(rng.choice([2, 3, 5, 17])), n, dtype=np.uint8)
        r = int(rng.integers(2, 40))
        return np.repeat(a[:(n + r - 1) // r], r)[:n].copy()
    if kind == 2:
        return rng.integers(0, int(rng.choice([2, 4, 16, 256])), n, dtype=np.uint8)
    return make_block(rng, n)


def rotate_paths(c, rng):
    """every round picks the decode paths anew: LZ4 in-wave parse / sequence index with 1 .. 64 walkers per block /
    the few-blocks path (batches it is not made for take the automatic choice) / automatic; zstd fused kernel / pipeline"""
    path = int(rng.choice([cc.LZ4_PATH_AUTO, cc.LZ4_PATH_RING, cc.LZ4_PATH_INDEXED, cc.LZ4_PATH_INDEXED, cc.LZ4_PATH_FEW_BLOCKS]))
    c.set_option(cc.OPT_LZ4_DECODE_PATH, path)
    c.set_option(cc.OPT_LZ4_INDEX_WALKERS, int(rng.choice([0, 1, 2, 4, 8, 16, 32, 64])) if path == cc.LZ4_PATH_INDEXED else 0)
    c.set_option(cc.OPT_LZ4_DECODE_WAVES, int(rng.choice([0, 1, 2])))  # indexed decoder: one wave per block / two
    c.set_option(cc.OPT_ZSTD_DECODE_PATH, int(rng.choice([0, 1, 2])))


def make_block(rng, n):
    out = np.empty(n, np.uint8)
    pos = 0
    vocab = [rng.integers(32, 127, int(rng.integers(3, 40)), dtype=np.uint8) for _ in range(int(rng.integers(4, 60)))]
    while pos < n:
        kind = int(rng.integers(0, 7))
        ln = int(min(n - pos, rng.integers(1, 4000) if kind != 3 else rng.integers(1, 40000)))
        if kind == 0:      # noise
            out[pos:pos + ln] = rng.integers(0, 256, ln, dtype=np.uint8)
        elif kind == 1:    # run
            out[pos:pos + ln] = rng.integers(0, 256)
        elif kind == 2 and pos > 8:   # repeat from anywhere earlier (any distance, overlapping allowed)
            src = int(rng.integers(0, pos))
            for i in range(ln):
                out[pos + i] = out[src + i]
        elif kind == 3 and pos > 8:   # long non-overlapping repeat
            src = int(rng.integers(0, pos))
            ln = min(ln, pos - src)
            out[pos:pos + ln] = out[src:src + ln]
        elif kind == 4:    # words
            i = 0
            while i < ln:
                w = vocab[int(rng.integers(0, len(vocab)))]
                k = min(len(w), ln - i)
                out[pos + i:pos + i + k] = w[:k]
                i += k
        elif kind == 5:    # low-entropy alphabet
            out[pos:pos + ln] = rng.choice(np.frombuffer(b"0123456789abcdef", np.uint8), ln)
        else:              # short period
            p = rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8)
            out[pos:pos + ln] = np.resize(p, ln)
        pos += ln
    return out


def mutate(rng, c):
    c = c.copy()
    for _ in range(int(rng.integers(1, 4))):
        k = int(rng.integers(0, 6))
        if len(c) < 8:
            break
        p = int(rng.integers(0, len(c)))
        if k == 0:
            c[p] ^= 1 << int(rng.integers(0, 8))
        elif k == 1:
            c[p] = rng.integers(0, 256)
        elif k == 2:
            c = c[:max(1, p)]
        elif k == 3:
            c = np.concatenate([c[:p], rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8), c[p:]])
        elif k == 4:
            q = min(len(c), p + int(rng.integers(1, 64)))
            c = np.concatenate([c[:p], c[q:]])
        else:   # early bytes: headers
            p = int(rng.integers(0, min(len(c), 24)))
            c[p] = rng.integers(0, 256)
    return c


def fuzz(budget, seed):
    """malformed streams: the device decoders' verdict and bytes equal the oracle's (which is pinned to the
    libraries by tests/test_oracle_golden.py), through the fused kernel (3 per call) and the pipeline (40)"""
    rng = np.random.default_rng(seed)
    stock = oracle_lib.StockLibs()
    ora = oracle_lib.Oracle()
    t_end = time.time() + budget
    rounds = bad_streams = 0
    with Codec(0) as c:
        while time.time() < t_end:
            rotate_paths(c, rng)
            B = int(rng.choice([4096, 20000, 131072, 300001]))
            base = [make_block(rng, B) for _ in range(4)]
            for method, name in ((METHOD_ZSTD, "zstd"), (METHOD_LZ4, "lz4")):
                valid = []
                for b in base:
                    valid.append(stock.zstd_compress(b, int(rng.choice([-3, 1, 3, 9]))) if method == METHOD_ZSTD
                                 else stock.lz4_compress(b, int(rng.choice([1, 20]))))
                for n in (3, 40):
                    items = [mutate(rng, valid[int(rng.integers(0, 4))]) if rng.random() < 0.85 else valid[int(rng.integers(0, 4))]
                             for _ in range(n)]
                    outs, st = c.decompress_blocks(method, items, B)
                    for i, m in enumerate(items):
                        r, exp = (ora.zstd_decompress(m, B, fill=0xA5) if method == METHOD_ZSTD else ora.lz4_decompress(m, B, fill=0xA5))
                        ok = (r == B)
                        if ok != (st[i] == 0) or (ok and not np.array_equal(outs[i], exp)):
                            os.makedirs("gpurun_out", exist_ok=True)
                            np.save("gpurun_out/fail_fuzz_%s.npy" % name, m)
                            raise AssertionError(("fuzz", name, seed, rounds, B, n, i, int(r), int(st[i])))
                        bad_streams += 0 if ok else 1
            rounds += 1
    print("fuzz ok: seed %d, %d rounds, %d rejected streams among them, %.0f s" % (seed, rounds, bad_streams, budget))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "fuzz":
        return fuzz(float(sys.argv[2]) if len(sys.argv) > 2 else 60.0, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    stock = oracle_lib.StockLibs()
    assert stock.lz4 is not None and stock.zstd is not None, "stock libraries needed"
    t_end = time.time() + budget
    rounds = checks = 0
    with Codec(0) as c:
        while time.time() < t_end:
            rotate_paths(c, rng)
            B = int(rng.choice([131072, 131072, 1 << 20, 65547, 70000, 20000, 300001, 9000, 200000]))
            n = int(rng.choice([3, 20, 40]))
            if rounds % 3 == 2:   # odd blocks: tiny ones, the size-class boundaries, few sequences, sparse alphabets, runs
                B = int(rng.choice([int(rng.integers(1, 600)), int(rng.integers(600, 20000)), 16384, 16385, 131073, 262145]))
                blocks = [odd_block(rng, B) for _ in range(n)]
            else:
                blocks = [make_block(rng, B) for _ in range(n)]
            accel = int(rng.choice([1, 1, 2, 9, 50, 300]))
            got = c.compress_blocks(METHOD_LZ4, accel, blocks)
            for i in range(n):
                exp = stock.lz4_compress(blocks[i], accel)
                if not np.array_equal(got[i], exp):
                    os.makedirs("gpurun_out", exist_ok=True)
                    np.save("gpurun_out/fail_lz4_enc_block.npy", blocks[i])
                    raise AssertionError(("lz4 enc", seed, rounds, B, accel, i, len(got[i]), len(exp)))
            outs, st = c.decompress_blocks(METHOD_LZ4, got, B)
            assert (st == 0).all() and all(np.array_equal(o, b) for o, b in zip(outs, blocks)), ("lz4 dec", seed, rounds, B)
            checks += 2 * n
            if True:   # every size class of libzstd's parameter tables has kernels for every level
                level = int(rng.choice([-5, -3, -1, 1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10] + ([11, 12] if B > 16384 else []) + ([13, 15] if B > 262144 else [])))
                gotz = c.compress_blocks(METHOD_ZSTD, level, blocks)
                if rounds % 4 == 0:   # an optimal-parser level (btopt / btultra / btultra2) on two blocks: seconds per block at 1 MiB
                    olevel = int(rng.choice([13, 16, 17, 19, 22] if B > 16384 else [11, 13, 17, 20, 22]))
                    goto = c.compress_blocks(METHOD_ZSTD, olevel, blocks[:2])
                    for i in range(min(n, 2)):
                        exp = stock.zstd_compress(blocks[i], olevel)
                        if not np.array_equal(goto[i], exp):
                            os.makedirs("gpurun_out", exist_ok=True)
                            np.save("gpurun_out/fail_zstd_opt_block.npy", blocks[i])
                            raise AssertionError(("zstd opt enc", seed, rounds, B, olevel, i, len(goto[i]), len(exp)))
                    checks += min(n, 2)
                for i in range(n):
                    exp = stock.zstd_compress(blocks[i], level)
                    if not np.array_equal(gotz[i], exp):
                        os.makedirs("gpurun_out", exist_ok=True)
                        np.save("gpurun_out/fail_zstd_enc_block.npy", blocks[i])
                        np.save("gpurun_out/fail_zstd_enc_got.npy", gotz[i])
                        raise AssertionError(("zstd enc", seed, rounds, B, level, i, len(gotz[i]), len(exp)))
                checks += n
            lvl = int(rng.choice([1, 3, 5, 9, 15, 19]))
            zs = [stock.zstd_compress(b, lvl) for b in blocks[:8 if lvl > 9 else n]]
            outs, st = c.decompress_blocks(METHOD_ZSTD, zs, B)
            assert (st == 0).all() and all(np.array_equal(o, b) for o, b in zip(outs, blocks)), ("zstd dec", seed, rounds, B, lvl)
            checks += len(zs)
            rounds += 1
    print("stress ok: seed %d, %d rounds, %d block checks in %.0f s" % (seed, rounds, checks, budget))


if __name__ == "__main__":
    main()
