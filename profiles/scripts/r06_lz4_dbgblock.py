# round 6 debugging aid: compress gpurun_out/fail_lz4_enc_block.npy on the GPU and print where its sequences part from stock liblz4's
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle_lib
from pg_cryogen_amd.codec import Codec, METHOD_LZ4

def seqs(s):
    s = bytes(s); i = 0; out = []; pos = 0
    while i < len(s):
        t = s[i]; i += 1
        ll = t >> 4
        if ll == 15:
            while True:
                b = s[i]; i += 1; ll += b
                if b != 255: break
        lit_at = i; i += ll; pos += ll
        if i >= len(s): out.append((pos - ll, ll, 0, 0)); break
        off = s[i] | (s[i + 1] << 8); i += 2
        ml = t & 15
        if ml == 15:
            while True:
                b = s[i]; i += 1; ml += b
                if b != 255: break
        ml += 4
        out.append((pos - ll, ll, off, ml)); pos += ml
    return out

b = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fail_lz4_enc_block.npy")
accel = int(sys.argv[2]) if len(sys.argv) > 2 else 1
stock = oracle_lib.StockLibs()
exp = stock.lz4_compress(b, accel)
with Codec(0) as c:
    got = c.compress_blocks(METHOD_LZ4, accel, [b])[0]
print("sizes", len(got), len(exp))
sg, se = seqs(got), seqs(exp)
for k, (x, y) in enumerate(zip(sg, se)):
    if x != y:
        print("first difference at sequence", k, "(anchor, literals, offset, match length): got", sg[k - 1:k + 3], "expected", se[k - 1:k + 3])
        a = x[0]
        print("bytes around", a, list(b[max(0, a - 16):a + 48]))
        break
else:
    print("same sequences", len(sg), len(se))
