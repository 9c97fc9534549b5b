"""Hand-built Zstandard frames for format corners libzstd's own encoder never emits (test infrastructure).

huf12_frame(): one compressed block whose literals use a Huffman table of log 12 (RFC 8878 4.2.1 allows
up to 12; libzstd's encoder stops at 11), direct 4-bit weights, 1 or 4 streams, no sequences.  The
decoders take a different path for such tables (two-level lookup in the batch pipeline)."""
import numpy as np

# 13 listed weights + 1 implied: 3 x 2^10 + 2^9 + ... + 2^0 + 2^0 = 4096 -> table log 12, code lengths 2..12
WEIGHTS = [11, 11, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 1]  # symbol 13's weight is implied by the others
LOG = 12


def _codes():
    """canonical codes as the decoder assigns them: weights ascending, symbols ascending inside a weight"""
    codes, nxt = {}, 0
    for w in range(1, LOG + 1):
        for s, ws in enumerate(WEIGHTS):
            if ws == w:
                span = 1 << (w - 1)
                codes[s] = (nxt // span, LOG + 1 - w)  # (value, nbits)
                nxt += span
    assert nxt == 1 << LOG
    return codes


def _stream(symbols, codes):
    acc, n = 0, 0
    for s in reversed(symbols):          # the decoder reads backwards: the first symbol sits on top
        v, nb = codes[int(s)]
        acc |= v << n
        n += nb
    acc |= 1 << n                        # end mark
    return acc.to_bytes((n + 8) // 8, "little")


def huf12_frame(n=700, streams=4, seed=1):
    rng = np.random.default_rng(seed)
    p = np.array([2.0 ** -(LOG + 1 - w) for w in WEIGHTS])
    lits = rng.choice(len(WEIGHTS), size=n, p=p / p.sum()).astype(np.uint8)
    lits[:14] = np.arange(14)            # every code, incl. both 12-bit ones, at least once
    codes = _codes()
    nw = len(WEIGHTS) - 1
    tree = bytes([127 + nw]) + bytes(((WEIGHTS[i] << 4) | (WEIGHTS[i + 1] if i + 1 < nw else 0)) for i in range(0, nw, 2))
    if streams == 1:
        body, fmt = _stream(lits, codes), 0
    else:
        seg = (n + 3) // 4
        parts = [_stream(lits[i * seg:(i + 1) * seg], codes) for i in range(4)]
        jump = b"".join(len(q).to_bytes(2, "little") for q in parts[:3])
        body, fmt = jump + b"".join(parts), 1
    csize = len(tree) + len(body)
    assert n < 1024 and csize < 1024
    lit_hdr = (2 | (fmt << 2) | (n << 4) | (csize << 14)).to_bytes(3, "little")
    block = lit_hdr + tree + body + b"\x00"          # literals, then "0 sequences"
    bh = ((len(block) << 3) | (2 << 1) | 1).to_bytes(3, "little")
    assert n >= 256
    frame = (0xFD2FB528).to_bytes(4, "little") + bytes([0x60]) + (n - 256).to_bytes(2, "little") + bh + block
    return np.frombuffer(frame, np.uint8).copy(), lits
