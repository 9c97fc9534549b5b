"""Round 6: is the index pass's speed class a property of the PROCESS or of where its buffers lie?
One process; the headline decode is timed, then buffers and workspace are freed and allocated again (same sizes, in another
order, with pads), and timed again.  If the class changes inside a process it is placement."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from pg_cryogen_amd import Codec, METHOD_LZ4, bound

B, n = 131072, 65536
stride = (bound(METHOD_LZ4, B) + 15) & ~15
codec = Codec(0)


def run(tag, pad_mb=0, order="rco"):
    pads = [codec.alloc(pad_mb << 20)] if pad_mb else []
    bufs = {}
    for k in order:
        bufs[k] = codec.alloc({"r": n * B, "c": n * stride, "o": n * B}[k])
    d_raw, d_comp, d_out = bufs["r"], bufs["c"], bufs["o"]
    d_sizes, d_off, d_status = codec.alloc(4 * n), codec.alloc(8 * n), codec.alloc(4 * n)
    d_off.upload(np.arange(n, dtype=np.uint64) * np.uint64(stride))
    codec.synth_batch(0, 0, n, B, 0, d_raw)
    codec.compress_batch(METHOD_LZ4, 1, d_raw, B, B, n, d_comp, stride, d_sizes, d_status)
    codec.sync()
    ts = []
    for i in range(24):
        codec.timer_start()
        codec.decompress_batch(METHOD_LZ4, d_comp, d_off, d_sizes, d_out, B, B, n, d_status)
        ts.append(codec.timer_stop())
    ts = sorted(ts[4:])
    print("%-28s median %.3f ms (min %.3f)  comp %#x out %#x" % (tag, ts[len(ts) // 2], ts[0], d_comp.ptr, d_out.ptr), flush=True)
    for b in list(bufs.values()) + [d_sizes, d_off, d_status] + pads:
        b.free()
    codec.trim()   # the workspace (sequence index rows) too


run("first")
run("again, same order")
run("again, same order")
run("order c r o", order="cro")
run("order o c r", order="orc")
run("pad 3 MiB in front", pad_mb=3)
run("pad 1 GiB in front", pad_mb=1024)
run("pad 17 MiB in front", pad_mb=17)
run("again, same order")
