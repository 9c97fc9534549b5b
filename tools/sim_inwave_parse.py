#!/usr/bin/env python3
"""Design aid (CPU only, uses the oracle as a data source): how fast do LZ4 token chains started at GUESSED positions
run into the true chain?  This sizes the in-wave parse of the single-pass LZ4 decoder:
lane i of a wave starts at byte i*seg of a window of 64*seg compressed bytes and walks; a lane's records become
true from the first position it shares with the true chain.

Prints, per distribution: the true hop length, and for guessed starts the bytes and hops walked before the merge.
"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from oracle_lib import Oracle  # noqa: E402


def next_token(c, p, n):
    """position of the token after the one at p (None: last sequence / cut / 255-run that leaves the fast path)"""
    if p >= n:
        return None
    t = c[p]
    ll = t >> 4
    q = p + 1
    if ll == 15:
        while True:
            if q >= n:
                return None
            b = c[q]
            q += 1
            ll += b
            if b != 255:
                break
    q += ll
    if q + 2 > n:
        return None
    q += 2
    if (t & 15) == 15:
        while True:
            if q >= n:
                return None
            b = c[q]
            q += 1
            if b != 255:
                break
    return q


def main():
    o = Oracle()
    B = 131072
    names = ["wide", "narrow", "int4", "random", "zeros"]
    for dist in (0, 1, 2):
        merge_bytes, merge_hops, true_hops = [], [], []
        per_seg = {32: [], 64: [], 128: []}
        for blk in range(4):
            raw = o.synth(1, blk, B, dist)
            c = o.lz4_compress(raw, 1)
            n = len(c)
            cl = c.tolist()
            truth = set()
            p = 0
            order = []
            while p is not None and p < n:
                truth.add(p)
                order.append(p)
                p = next_token(cl, p, n)
            true_hops += list(np.diff(order))
            rng = np.random.default_rng(blk)
            for g in rng.integers(0, max(1, n - 600), 2000):
                p = int(g)
                hops = 0
                while p is not None and p < n and p not in truth and hops < 400:
                    p = next_token(cl, p, n)
                    hops += 1
                if p is not None and p < n and hops < 400:
                    merge_bytes.append(p - int(g))
                    merge_hops.append(hops)
                else:
                    merge_bytes.append(10 ** 6)
                    merge_hops.append(400)
            # lockstep cost: hops a wave needs so that every lane has passed checkpoint (i + r) * seg
            for seg in per_seg:
                for w0 in range(0, max(1, n - 64 * seg - 600), 64 * seg * 4):
                    worst = 0
                    for i in range(64):
                        p = order[np.searchsorted(order, w0)] if i == 0 else w0 + i * seg
                        lim = w0 + (i + 3) * seg
                        h = 0
                        while p is not None and p < lim and h < 400:
                            p = next_token(cl, p, n)
                            h += 1
                        worst = max(worst, h)
                    per_seg[seg].append(worst)
        mb = np.array(merge_bytes)
        mh = np.array(merge_hops)
        th = np.array(true_hops)
        print(f"{names[dist]}: csize {n}, sequences {len(order)}, true hop mean {th.mean():.1f} median {np.median(th):.0f}")
        print(f"   guessed start -> merge: bytes median {np.median(mb):.0f} p90 {np.percentile(mb, 90):.0f} p99 {np.percentile(mb, 99):.0f}"
              f" ; hops median {np.median(mh):.0f} p90 {np.percentile(mh, 90):.0f} p99 {np.percentile(mh, 99):.0f}")
        for seg, v in per_seg.items():
            v = np.array(v)
            print(f"   seg {seg}: hops until every lane passed 3 segments: mean {v.mean():.1f} max {v.max()}")


if __name__ == "__main__":
    main()
