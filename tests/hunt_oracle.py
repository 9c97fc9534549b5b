"""Differential hunt on the CPU (not collected by pytest: `python tests/hunt_oracle.py [seconds] [seed] [zstd|lz4]`): the zstd encoder
oracle against the live libzstd.so.1 at random levels (-5 .. 22) -- or the LZ4 encoder oracle against liblz4.so.1 at random
accelerations -- on random blocks of every kind the GPU soak found trouble with --
tiny and small blocks, the size-class boundaries, structured blocks, periodic blocks of a few sequences, sparse alphabets and
runs.  Run it under AddressSanitizer too (tests/run_sanitized.sh does, briefly): that is how the unbounded Huffman write of
the restatement was found."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_lib


def make_case(rng, make_block, few_sequence_blocks):
    kind = int(rng.integers(0, 6))
    if kind == 0:
        B = int(rng.integers(1, 600))
    elif kind == 1:
        B = int(rng.integers(600, 20000))
    elif kind == 2:
        B = int(rng.choice([16384, 16385, 131072, 131073, 262144, 262145]))
    elif kind == 3:
        B = int(rng.integers(20000, 300000))
    else:
        B = int(rng.integers(100, 12000))
    if kind == 4:
        return few_sequence_blocks(int(rng.integers(0, 1 << 30)), 1)[0]
    if kind == 5:   # sparse alphabets, runs
        a = rng.integers(0, int(rng.choice([2, 3, 5, 17])), B, dtype=np.uint8)
        if rng.random() < 0.5:
            r = int(rng.integers(2, 40))
            a = np.repeat(a[:max(1, B // r)], r)[:B]
        return a.copy()
    if B < 4096 and rng.random() < 0.5:
        return rng.integers(0, int(rng.choice([2, 4, 16, 256])), B, dtype=np.uint8)
    return make_block(rng, B)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lz4 = len(sys.argv) > 3 and sys.argv[3] == "lz4"
    from stress_gpu import make_block
    from test_oracle_golden import few_sequence_blocks
    o = oracle_lib.Oracle()
    st = oracle_lib.StockLibs()
    assert (st.lz4 if lz4 else st.zstd) is not None, "the stock library is needed"
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n = bad = 0
    while time.time() < t_end:
        blk = make_case(rng, make_block, few_sequence_blocks)
        if len(blk) == 0:
            continue
        params = rng.choice([0, 1, 2, 3, 7, 9, 50, 300, 65537], 4, replace=False) if lz4 else rng.choice(np.arange(-5, 23), 4, replace=False)
        for lvl in params:
            lvl = int(lvl)
            n += 1
            same = (np.array_equal(st.lz4_compress(blk, lvl), o.lz4_compress(blk, lvl)) if lz4
                    else np.array_equal(st.zstd_compress(blk, lvl), o.zstd_compress(blk, lvl)))
            if not same:
                bad += 1
                np.save("hunt_fail_%d_%d_level%d.npy" % (seed, n, lvl), blk)
                print("MISMATCH seed", seed, "case", n, "len", len(blk), "level", lvl, flush=True)
    print("hunt %s: seed %d, %d cases, %d mismatches" % ("ok" if bad == 0 else "FAILED", seed, n, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
