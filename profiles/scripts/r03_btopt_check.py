"""zstd encode at the optimal-parser levels (btopt / btultra / btultra2): the kernel against the oracle and the stock library,
with the time per call.  Usage: r03_btopt_check.py [quick]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle_lib
from stress_gpu import make_block
from pg_cryogen_amd.codec import Codec, METHOD_ZSTD

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
o = oracle_lib.Oracle(); st = oracle_lib.StockLibs(); codec = Codec()
cases = [(1500, range(11, 23), 4), (16384, range(11, 23), 4), (20000, range(13, 23), 4), (131072, (13, 14, 16, 17, 19, 22), 4)]
if not quick:
    cases += [(200000, (13, 16, 19, 22), 2), (262145, (16, 17, 19), 2), (1 << 20, (16, 18, 19, 22), 2)]
bad = 0
for B, levels, nb in cases:
    rng = np.random.default_rng(B)
    blocks = [make_block(rng, B)] + [o.synth(7, i, B, i % 5) if B >= 4096 else rng.integers(0, 4, B, dtype=np.uint8) for i in range(nb - 1)]
    for lvl in levels:
        t0 = time.time()
        got = codec.compress_blocks(METHOD_ZSTD, lvl, blocks)
        dt = time.time() - t0
        ok_o = all(np.array_equal(g, o.zstd_compress(b, lvl)) for g, b in zip(got, blocks)) if B <= (1 << 20) else None
        ok_s = all(np.array_equal(g, st.zstd_compress(b, lvl)) for g, b in zip(got, blocks))
        outs, stt = codec.decompress_blocks(METHOD_ZSTD, got, B)
        rt = bool((stt == 0).all() and all(np.array_equal(x, b) for x, b in zip(outs, blocks)))
        bad += (ok_o is False) + (not ok_s) + (not rt)
        print("B %7d level %2d: %d blocks in %6.2f s; equals oracle %s, stock libzstd %s, decodes %s" % (B, lvl, len(blocks), dt, ok_o, ok_s, rt), flush=True)
print("mismatches:", bad)
