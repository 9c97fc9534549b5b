#!/usr/bin/env python3
"""PCIe-inclusive rate of the C host-buffer batch API (cryo_codec_{compress,decompress}_blocks: pageable host
memory in, pageable host memory out, the calls host/staging.c and host/cache.c make), for DESIGN.md section 5.
Not the bench metric."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pg_cryogen_amd import Codec, METHOD_LZ4, METHOD_ZSTD, bound

def run(c, B, n, method, param, name):
    L = c.L
    d = c.alloc(n * B)
    c.synth_batch(0, 0, n, B, 0, d)
    c.sync()
    raw = d.download()
    d.free()
    cap = bound(method, B)
    comp = np.zeros(n * cap, np.uint8)
    sizes = np.zeros(n, np.uint32)
    out = np.zeros(n * B, np.uint8)
    st = np.zeros(n, np.int32)
    def enc():
        rc = L.cryo_codec_compress_blocks(c.h, method, param, raw.ctypes.data, B, n, comp.ctypes.data, cap, sizes.ctypes.data)
        assert rc == 0, rc
    ptrs = (C.c_void_p * n)(*[comp.ctypes.data + i * cap for i in range(n)])
    def dec():
        rc = L.cryo_codec_decompress_blocks(c.h, method, ptrs, sizes.ctypes.data, n, out.ctypes.data, B, st.ctypes.data)
        assert rc == 0, rc
    enc(); dec()                       # warm up: buffers grow, pages get touched
    te, td = [], []
    for _ in range(REPS):
        t0 = time.perf_counter(); enc(); t1 = time.perf_counter(); dec(); t2 = time.perf_counter()
        te.append(n * B / (t1 - t0) / 1e9); td.append(n * B / (t2 - t1) / 1e9)
    assert (st == 0).all() and np.array_equal(out, raw)
    rng = lambda v: "%.2f (%.2f-%.2f)" % (sorted(v)[len(v) // 2], min(v), max(v))
    print("%s host API, %d x %d KiB: compress %s GB/s, decompress %s GB/s; median (min-max) of %d calls, uncompressed bytes, "
          "PCIe and host copies included" % (name, n, B // 1024, rng(te), rng(td), REPS), flush=True)


def run_multi(B, n, method, param, name, G):
    """the same through the multi-GPU dispatcher with G handles (on a 1-GPU box they share device 0): block i -> handle i mod G"""
    from pg_cryogen_amd import codec as cc
    L = cc.lib()
    h = C.c_void_p()
    devs = (C.c_int * G)(*([0] * G))
    assert L.cryo_multi_open(devs, G, C.byref(h)) == 0
    with Codec(0) as c:
        d = c.alloc(n * B)
        c.synth_batch(0, 0, n, B, 0, d)
        c.sync()
        raw = d.download()
        d.free()
    cap = bound(method, B)
    comp = np.zeros(n * cap, np.uint8)
    sizes = np.zeros(n, np.uint32)
    out = np.zeros(n * B, np.uint8)
    st = np.zeros(n, np.int32)
    ptrs = (C.c_void_p * n)(*[comp.ctypes.data + i * cap for i in range(n)])
    def enc():
        assert L.cryo_multi_compress_blocks(h, method, param, raw.ctypes.data, B, n, comp.ctypes.data, cap, sizes.ctypes.data) == 0
    def dec():
        assert L.cryo_multi_decompress_blocks(h, method, ptrs, sizes.ctypes.data, n, out.ctypes.data, B, st.ctypes.data) == 0
    enc(); dec()
    te, td = [], []
    for _ in range(REPS):
        t0 = time.perf_counter(); enc(); t1 = time.perf_counter(); dec(); t2 = time.perf_counter()
        te.append(n * B / (t1 - t0) / 1e9); td.append(n * B / (t2 - t1) / 1e9)
    assert (st == 0).all() and np.array_equal(out, raw)
    L.cryo_multi_close(h)
    rng = lambda v: "%.2f (%.2f-%.2f)" % (sorted(v)[len(v) // 2], min(v), max(v))
    print("%s host API through cryo_multi_* with %d handles on one GPU, %d x %d KiB: compress %s GB/s, decompress %s GB/s"
          % (name, G, n, B // 1024, rng(te), rng(td)), flush=True)


def topology():
    """where the GPU hangs and where this process may run: the staging copies and the pinned buffers should be on the GPU's
    NUMA node (a decompress call moves ~1.8 x its output across PCIe; the host copies run at memory speed)"""
    import glob
    for dev in sorted(glob.glob("/sys/bus/pci/devices/*")):
        try:
            cls = open(dev + "/class").read()
            if open(dev + "/vendor").read().strip() != "0x1002" or not (cls.startswith("0x03") or cls.startswith("0x12")):
                continue
            print("[topology] %s numa_node %s local_cpulist %s" % (os.path.basename(dev), open(dev + "/numa_node").read().strip(),
                                                                     open(dev + "/local_cpulist").read().strip()))
        except OSError:
            pass
    print("[topology] process affinity: %d cpus (%s ...), cgroup cpu.max: %s" % (
        len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8],
        (open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a")), flush=True)

def single(c, B, method, param, name):
    """the reference's own call shape: ONE block per call (pg_cryogen.c:726, cache.c:178), next to the stock library
    on one host core"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    stock = oracle_lib.StockLibs()
    d = c.alloc(B)
    c.synth_batch(0, 5, 1, B, 0, d)
    c.sync()
    raw = d.download()
    d.free()
    comp = c.compress_block(method, param, raw)
    for _ in range(3):
        out = c.decompress_block(method, comp, B)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); out = c.decompress_block(method, comp, B); ts.append(time.perf_counter() - t0)
    assert np.array_equal(out, raw)
    te = []
    for _ in range(5):
        t0 = time.perf_counter(); c.compress_block(method, param, raw); te.append(time.perf_counter() - t0)
    cpu_d = cpu_e = float("nan")
    if (stock.lz4 if method == METHOD_LZ4 else stock.zstd) is not None:
        dec = stock.lz4_decompress if method == METHOD_LZ4 else stock.zstd_decompress
        enc = stock.lz4_compress if method == METHOD_LZ4 else stock.zstd_compress
        td = []
        for _ in range(20):
            t0 = time.perf_counter(); dec(comp, B); td.append(time.perf_counter() - t0)
        tc = []
        for _ in range(5):
            t0 = time.perf_counter(); enc(raw, param); tc.append(time.perf_counter() - t0)
        cpu_d, cpu_e = sorted(td)[len(td) // 2], sorted(tc)[len(tc) // 2]
    print("%s ONE block of %d KiB per call (cryo_codec_{de,}compress_block, PCIe included): decompress %.3f ms, compress %.3f ms; "
          "stock library on one host core (through ctypes): decompress %.3f ms, compress %.3f ms"
          % (name, B // 1024, sorted(ts)[len(ts) // 2] * 1e3, sorted(te)[len(te) // 2] * 1e3, cpu_d * 1e3, cpu_e * 1e3))


REPS = int(os.environ.get("HOST_API_REPS", "7"))
topology()
if os.environ.get("HOST_API_CPUS"):   # e.g. the GPU's local_cpulist: "0-63"
    want = set()
    for part in os.environ["HOST_API_CPUS"].split(","):
        a, _, b = part.partition("-")
        want |= set(range(int(a), int(b or a) + 1))
    os.sched_setaffinity(0, want & os.sched_getaffinity(0))
    print("[topology] restricted to %d cpus" % len(os.sched_getaffinity(0)), flush=True)
if os.environ.get("HOST_API_MULTI"):
    for G in (1, 2):
        run_multi(131072, 4096, METHOD_LZ4, 1, "lz4", G)
        run_multi(131072, 4096, METHOD_ZSTD, 1, "zstd", G)
        run_multi(1 << 20, 512, METHOD_LZ4, 1, "lz4", G)
    sys.exit(0)
with Codec(0) as c:
    from pg_cryogen_amd import codec as cc
    if os.environ.get("HOST_API_NUMA") == "0":
        c.set_option(cc.OPT_NUMA_LOCAL, 0)
    print("[topology] CRYO_OPT_NUMA_LOCAL in effect: %d" % c.get_option(cc.OPT_NUMA_LOCAL), flush=True)
    for B, n in ((131072, 4096), (1 << 20, 16), (1 << 20, 512)):
        run(c, B, n, METHOD_LZ4, 1, "lz4")
        run(c, B, n, METHOD_ZSTD, 1, "zstd")
    for B in (131072, 1 << 20):
        single(c, B, METHOD_LZ4, 1, "lz4")
        single(c, B, METHOD_ZSTD, 1, "zstd")
