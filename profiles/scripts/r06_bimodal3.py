"""Round 6: which buffer's placement decides the index pass's speed class?  One process, the headline decode; between timings
exactly one thing is freed and allocated again: the handle's workspace (the sequence-index rows), the compressed input, the output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from pg_cryogen_amd import Codec, METHOD_LZ4, bound

B, n = 131072, 65536
stride = (bound(METHOD_LZ4, B) + 15) & ~15
codec = Codec(0)
d_raw = codec.alloc(n * B); d_comp = codec.alloc(n * stride); d_out = codec.alloc(n * B)
d_sizes, d_off, d_status = codec.alloc(4 * n), codec.alloc(8 * n), codec.alloc(4 * n)
d_off.upload(np.arange(n, dtype=np.uint64) * np.uint64(stride))
codec.synth_batch(0, 0, n, B, 0, d_raw)


def enc():
    codec.compress_batch(METHOD_LZ4, 1, d_raw, B, B, n, d_comp, stride, d_sizes, d_status); codec.sync()


def t(tag):
    ts = []
    for i in range(20):
        codec.timer_start()
        codec.decompress_batch(METHOD_LZ4, d_comp, d_off, d_sizes, d_out, B, B, n, d_status)
        ts.append(codec.timer_stop())
    ts = sorted(ts[4:])
    print("%-34s median %.3f ms" % (tag, ts[len(ts) // 2]), flush=True)


enc(); t("start")
for i in range(6):
    codec.trim(); t("workspace allocated again (%d)" % i)
for i in range(5):
    d_comp.free(); d_comp = codec.alloc(n * stride); enc(); t("compressed input again (%d)" % i)
for i in range(5):
    d_out.free(); d_out = codec.alloc(n * B); t("output again (%d)" % i)
