#!/usr/bin/env python3
"""bench.py -- benchmark of the cryo-block codec hot path on MI355X.

Metric (BASELINE.json): uncompressed GB/s over a batch of synthetic 128 KiB cryo blocks resident in HBM,
bit-exact against liblz4 / libzstd.

  --workload lz4_decode (default, headline)   LZ4 decompress.  N = 1: BASELINE configs[1], 65 536 blocks.
                                              N > 1: configs[3], 1 Mi blocks over 8 GPUs = 131 072 blocks
                                              per GPU (the same per-GPU share at N = 2, 4: weak scaling),
                                              block i of the job on rank i mod N, no collective.
  --workload zstd_decode                      same shape, zstd level --level
  --workload lz4 | zstd                       compress + decompress of the batch (configs[2] shape)
  --workload mixed                            configs[4]: even blocks zstd level 22, odd blocks lz4 acceleration 50
                                              (both by the GPU encoders, untimed setup); decode of both

A "step" = one pass of the workload over the rank's whole batch.  Setup (untimed): blocks are generated on the
device, compressed on the device by the bit-exact HIP encoder, and a strided sample is verified against the CPU
oracle; after the timed region every decoded block is compared on the device with its original.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (child
processes, before anything touches the GPU) and relays rank 0's line; under torch.distributed.run it uses the
ranks it is given.  Prints ONE JSON line on rank 0 (see the driver contract in the task).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
CONFIG4_TOTAL_BLOCKS = 1 << 20  # BASELINE.json configs[3]: 1 M x 128 KB blocks over 8 GPUs


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="default: a timed region of about 2 s")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=0, help="cryo blocks per GPU (default: 65536 at N=1, 131072 at N>1, 16384 for mixed)")
    ap.add_argument("--block-size", type=int, default=131072)
    ap.add_argument("--dist", default="wide", choices=["wide", "narrow", "int4", "random", "zeros"])
    ap.add_argument("--accel", type=int, default=1, help="lz4 acceleration used to produce the inputs")
    ap.add_argument("--cpu-blocks", type=int, default=4096, help="distinct blocks of the cpu_baseline sample (SURVEY.md 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="diagnostic ablation runs only: result is not valid")
    ap.add_argument("--workload", default="lz4_decode", choices=["lz4_decode", "zstd_decode", "zstd", "lz4", "mixed"])
    ap.add_argument("--level", type=int, default=1, help="zstd level")
    ap.add_argument("--lz4-path", type=int, default=0, help="diagnostic: CRYO_OPT_LZ4_DECODE_PATH (0 auto, 1 in-wave parse, 2 indexed)")
    ap.add_argument("--flush", choices=["auto", "on", "off"], default="auto",
                    help="decode workloads: sweep a 512 MiB scratch buffer before every timed step, so that a small batch does not "
                         "find its input and output in the 256 MiB Infinity Cache (SURVEY.md 8d); auto = on when the batch's "
                         "buffers are below 2 GiB.  With the flush on, value is the rate of the timed steps (HIP events), the "
                         "flushes lie outside them")
    ap.add_argument("--ref-gbps", type=float, default=0.0,
                    help="one-GPU rate of the same per-GPU share: per_gpu_efficiency_vs_n1 = slowest rank's rate / this (N > 1)")
    ap.add_argument("--lz4-walkers", type=int, default=0, help="diagnostic: CRYO_OPT_LZ4_INDEX_WALKERS (0 auto)")
    ap.add_argument("--lz4-waves", type=int, default=0, help="diagnostic: CRYO_OPT_LZ4_DECODE_WAVES (0 auto, 1 one wave per block, 2 two)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks here, before any GPU / torch.cuda call in this process
# ------------------------------------------------------------------------------------------------
def self_launch(a):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    sys.exit(rc)


# ------------------------------------------------------------------------------------------------
# cpu_baseline: the library the reference links, on this machine's host cores
# ------------------------------------------------------------------------------------------------
def physical_cores():
    """One CPU number per physical core this process may run on (all PHYSICAL cores, pinned): the first
    hardware thread of every distinct sibling set in sysfs, restricted to the process's affinity mask."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, cores = set(), []
    for c in allowed:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                sib = f.read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            cores.append(c)
    return cores or allowed


def cpu_quota():
    """CPUs' worth of time the container's cgroup grants this process (cpu.max / cfs quota), or None: a 256-thread host
    whose container is given 16 CPUs runs 128 pinned threads at 16 cores' pace -- and at more in the one run in nine the
    quota bursts (profiles/r03_cpu_baseline_repeat.txt: 85-122 GB/s, once 301)."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(q) // int(p))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // p)
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(ora, np, method, encode, param, blobs, sizes, B, reps=5):
    """One pass over the sample's distinct blocks, median of `reps`: stock liblz4 / libzstd called
    exactly as reference compression.c:70-72,84,102-104,116 calls them, on 1 pinned thread (`value`) and on one pinned
    thread per physical core -- as many of them as the container's CPU quota lets run at once -- (`all_cores_value`,
    `cores_used`); the oracle port's 1-thread rate beside it."""
    fn = ora.L.cryo_oracle_cpu_pass_bench
    fn.restype = ctypes.c_double
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                   ctypes.c_char_p, ctypes.c_size_t]
    n = len(sizes)
    offs = np.zeros(n, np.uint64)
    pos = 0
    for i in range(n):
        offs[i] = pos
        pos += (int(sizes[i]) + 63) & ~63
    packed = np.zeros(pos + 64, np.uint8)
    for i in range(n):
        packed[int(offs[i]):int(offs[i]) + int(sizes[i])] = blobs[i]
    sz = np.asarray(sizes, np.uint32)
    ver = ctypes.create_string_buffer(64)

    cores = physical_cores()
    quota = cpu_quota()
    if quota is not None and quota < len(cores):
        cores = cores[:quota]   # no more threads than the container may run at once
    pins = (ctypes.c_int * len(cores))(*cores)

    def run(stock, threads, r):
        # one thread: one pass per timed region; T threads: T/8 passes, so that a region is tens of milliseconds
        inner = 1 if threads == 1 else max(1, threads // 8)
        return fn(method, 1 if encode else 0, stock, param, packed.ctypes.data, offs.ctypes.data, sz.ctypes.data, n, B, threads,
                  pins, r, inner, None, ver, 64)
    T = len(cores)
    lib = "liblz4" if method == 0 else "libzstd"
    call = {(0, False): "LZ4_decompress_safe(src,dst,csize,B)", (0, True): "LZ4_compress_fast(src,dst,B,bound,accel)",
            (1, False): "ZSTD_decompress(dst,B,src,csize)", (1, True): "ZSTD_compress(dst,bound,src,B,level)"}[(method, encode)]
    one = run(1, 1, reps)
    port = run(0, 1, max(1, reps // 2))
    res = {"unit": "GB/s", "cores": 1,
           "sample": "%d distinct blocks (%.0f MiB uncompressed), one pass per timed region on 1 thread (threads/8 passes with all cores), median of %d" % (n, n * B / 2**20, reps)}
    if one > 0:
        allc = run(1, T, reps)
        res.update({"value": round(one, 3), "kind": "reference", "library": "%s %s via dlopen (what the reference links, Makefile:5); %s"
                    % (lib, ver.value.decode(), call), "all_cores_value": round(allc, 3), "threads": T, "cores_used": T,
                    "pinning": "one thread per physical core (first hardware thread of each sibling set), pthread_setaffinity_np; every thread decodes from its own first-touched copy of its share of the input (NUMA-local)",
                    "hardware_threads": os.cpu_count(), "physical_cores": len(physical_cores()), "cgroup_cpu_quota": quota, "port_value": round(port, 3)})
    else:   # the stock library is not on this machine: the restatement is all there is
        res.update({"value": round(port, 3), "kind": "port", "library": "oracle/ restatement (stock %s not loadable here)" % lib})
    return res


def lookup_traffic(mname, n, B, dist, param):
    """HBM bytes per launch of this workload and where they come from: the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    passes of the same command (profiles/collect.sh, corrected by profiles/summarize.py as MI355X_MICROARCH.md prescribes),
    committed as profiles/*_hbm_traffic.json.  Counters cannot be read inside a timed run; a workload without a committed
    pass reports null."""
    try:
        for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if fn.endswith("_hbm_traffic.json"):
                t = json.load(open(os.path.join(ROOT, "profiles", fn)))
                wl = t.get("workload", {})
                if (wl.get("method", "lz4"), wl.get("blocks_per_gpu"), wl.get("block_size"), wl.get("distribution"),
                        wl.get("param", wl.get("lz4_acceleration"))) == (mname, n, B, dist, param):
                    return t["traffic_bytes_per_launch"], "profiles/%s (separate --pmc passes of this command, not this run)" % fn
    except OSError:
        pass
    return None, None


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)   # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not a.blocks:
        a.blocks = 16384 if a.workload == "mixed" else (65536 if world == 1 else CONFIG4_TOTAL_BLOCKS // 8)

    import numpy as np
    import torch  # plumbing only: process group + device sync; loaded first so ONE HIP runtime is shared
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    have_cuda = torch.cuda.is_available()
    ndev = torch.cuda.device_count() if have_cuda else 0
    # one process per GPU; if a node exposes fewer devices than ranks (functional tests of the N>1
    # path on a 1-GPU box) ranks share devices round-robin
    dev = local_rank % ndev if ndev else local_rank
    if have_cuda:
        torch.cuda.set_device(dev)

    from pg_cryogen_amd import Codec, METHOD_LZ4, METHOD_ZSTD, bound
    from pg_cryogen_amd.codec import DIST_NAMES
    import oracle_lib

    def barrier():
        if have_cuda:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    B, n = a.block_size, a.blocks
    dist_id = DIST_NAMES.index(a.dist)
    codec = Codec(dev)
    if a.lz4_path or a.lz4_walkers or a.lz4_waves:
        from pg_cryogen_amd import codec as cc
        codec.set_option(cc.OPT_LZ4_DECODE_PATH, a.lz4_path)
        codec.set_option(cc.OPT_LZ4_INDEX_WALKERS, a.lz4_walkers)
        codec.set_option(cc.OPT_LZ4_DECODE_WAVES, a.lz4_waves)
    ora = oracle_lib.Oracle()
    job_block = lambda k: rank + k * world   # block i of the job lives on rank i mod N (pg_cryogen_amd/shard.py)
    # cpu_baseline: rank 0, at every world size, after the timed region (north_star: the host's stock libraries timed "in the
    # same run" at 1, 2, 4 and 8 GPUs).  The other ranks are past their last barrier by then and only free their buffers.
    want_cpu = rank == 0 and not a.no_cpu_baseline
    bufs = []

    def alloc(nbytes):
        b = codec.alloc(nbytes)
        bufs.append(b)
        return b

    rank_times = []   # every rank's own wall time of the timed region (filled by max_over_ranks)

    def max_over_ranks(x):
        """MAX over the ranks (the contract's `value` uses it); also keeps every rank's own time: one aggregate hides a
        straggler GPU, and north_star's ">= 0.9 x per-GPU efficiency at 8 GPUs" is a statement about the slowest rank."""
        t = torch.tensor([x], dtype=torch.float64)
        del rank_times[:]
        if world > 1:
            g = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(g, t)
            rank_times.extend(float(v[0]) for v in g)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        else:
            rank_times.append(x)
        return float(t[0])

    def per_rank_fields(steps, per_rank_bytes):
        """per_rank_ms_per_step, slowest_rank, the slowest rank's own rate and -- for N > 1 -- its ratio to the one-GPU rate of
        the same per-GPU share (--ref-gbps, or the committed profiles/r05n_lz4_decode_bench.json for the configs[3] share)."""
        ms = [round(t / steps * 1e3, 4) for t in rank_times]
        slow = max(range(len(ms)), key=lambda i: ms[i])
        rate = per_rank_bytes * steps / rank_times[slow] / 1e9
        out = {"per_rank_ms_per_step": ms, "slowest_rank": slow, "slowest_rank_GBps": round(rate, 2)}
        ref = a.ref_gbps
        if not ref and world > 1 and a.workload == "lz4_decode" and n == CONFIG4_TOTAL_BLOCKS // 8 and B == 131072 and a.dist == "wide":
            try:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05n_lz4_decode_bench.json")) as f:
                    ref = float(json.load(f)["value"])
            except (OSError, ValueError, KeyError):
                ref = 0.0
        if ref and world > 1:
            out["per_gpu_efficiency_vs_n1"] = round(rate / ref, 4)
            out["n1_reference_GBps"] = ref
        return out

    def oracle_encode(method, param, k):
        raw = ora.synth(0, job_block(k), B, dist_id)
        return ora.zstd_compress(raw, param) if method == METHOD_ZSTD else ora.lz4_compress(raw, param)

    d_raw = alloc(n * B)
    ostride = B + int(os.environ.get("CRYO_BENCH_OUT_PAD", "0"))   # pad: layout experiments (decode workloads)
    d_out = alloc(n * ostride)
    d_status, d_mis = alloc(4 * n), alloc(8)
    codec.synth_batch(0, rank, n, B, dist_id, d_raw, block_step=world)
    ncpu = min(a.cpu_blocks, n)
    cpu_idx = list(range(0, n, max(1, n // ncpu)))[:ncpu]

    base = {"n_gpus": world, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic"}

    def verify_all():
        st = d_status.download(dtype=np.int32)
        assert a.no_verify or (st == 0).all(), "decode status"
        d_mis.memset(0)
        codec.compare_batch(d_raw, B, d_out, ostride if a.workload.endswith("_decode") else B, B, n, d_mis)
        codec.sync()
        mismatch = int(d_mis.download(dtype=np.uint64)[0])
        assert a.no_verify or mismatch == 0, "decoded blocks differ from originals: %d" % mismatch

    # =============================== decode workloads (headline) ===============================
    if a.workload.endswith("_decode"):
        is_lz4 = a.workload == "lz4_decode"
        method, param, mname = (METHOD_LZ4, a.accel, "lz4") if is_lz4 else (METHOD_ZSTD, a.level, "zstd")
        stride = ((bound(method, B) + 15) & ~15) + int(os.environ.get("CRYO_BENCH_STRIDE_PAD", "0"))   # pad: layout experiments
        d_comp, d_sizes, d_off = alloc(n * stride), alloc(4 * n), alloc(8 * n)
        codec.timer_start()
        codec.compress_batch(method, param, d_raw, B, B, n, d_comp, stride, d_sizes, d_status)
        enc_ms = codec.timer_stop()
        assert (d_status.download(dtype=np.int32) == 0).all(), "encode status"
        sizes = d_sizes.download(dtype=np.uint32)
        d_off.upload(np.arange(n, dtype=np.uint64) * np.uint64(stride))
        if os.environ.get("CRYO_BENCH_SAME_BLOCK"):   # diagnostic (with --no-verify): every block decodes block 0's stream -- the
            d_off.upload(np.zeros(n, dtype=np.uint64))   # input and its index rows come from cache, the work per block is unchanged
            d_sizes.upload(np.full(n, sizes[0], dtype=np.uint32))
        comp_bytes = int(sizes.astype(np.uint64).sum())
        sample_idx = sorted(set(list(range(0, n, max(1, n // 16)))[:16] + [n - 1]))
        for i in sample_idx:   # spot check vs the CPU oracle
            c = d_comp.download(int(sizes[i]), offset=i * stride)
            assert np.array_equal(c, oracle_encode(method, param, i)), "device encode differs from oracle at block %d" % i
        cpu_comps = [d_comp.download(int(sizes[i]), offset=i * stride) for i in cpu_idx] if want_cpu else []

        def step():
            codec.decompress_batch(method, d_comp, d_off, d_sizes, d_out, ostride, B, n, d_status)
        for _ in range(a.warmup):
            step()
        codec.sync()
        barrier()
        t0 = time.perf_counter()
        kernel_ms = []
        # cache flush: re-decoding the same small batch would read its input from, and write its output to, the
        # 256 MiB Infinity Cache; a 512 MiB memset between the steps (outside the timed launches) evicts both
        flush = a.flush == "on" or (a.flush == "auto" and n * (stride + ostride) < (2 << 30))
        d_flush = alloc(512 << 20) if flush else None
        for _ in range(a.steps):
            if flush:
                d_flush.memset(0x5A)
                codec.sync()
            codec.timer_start()          # HIP events on the codec's own stream
            step()
            kernel_ms.append(codec.timer_stop())
        codec.sync()
        barrier()
        elapsed = max_over_ranks((sum(kernel_ms) * 1e-3) if flush else (time.perf_counter() - t0))
        verify_all()
        if rank == 0 and os.environ.get("CRYO_BENCH_TRACE"):   # diagnostic: the per-step times, ten per line
            print("[bench trace] device pointers: comp %#x out %#x raw %#x" % tuple(int(getattr(x, "ptr", 0) or 0) for x in (d_comp, d_out, d_raw)), file=sys.stderr)
            for i in range(0, len(kernel_ms), 10):
                print("[bench trace] steps %3d..: %s" % (i, " ".join("%.2f" % x for x in kernel_ms[i:i + 10])), file=sys.stderr)
        if rank == 0:
            avg_ms = float(np.mean(kernel_ms))
            algo_bytes = comp_bytes + n * B        # per decode call: compressed bytes read + B written per block
            achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
            value = n * world * B * a.steps / elapsed / 1e9
            out = dict(base, metric="%s_decompress_uncompressed_GBps" % mname, value=round(value, 2), unit="GB/s",
                       ms_per_step=round(elapsed / a.steps * 1e3, 4), per_gpu_GBps=round(value / world, 2))
            out.update(per_rank_fields(a.steps, n * B))
            out["config"] = {"workload": "%s decompress %d x %d KiB synthetic cryo blocks per GPU" % (mname.upper(), n, B // 1024)
                             + (" (BASELINE configs[1])" if (world == 1 and n == 65536 and is_lz4) else "")
                             + (" (BASELINE configs[3]: 1 Mi blocks over 8 GPUs)" if (n * world == CONFIG4_TOTAL_BLOCKS and is_lz4) else ""),
                             "distribution": a.dist, "method": mname, "param": param, "block_size": B,
                             "blocks_per_gpu": n, "total_blocks": n * world, "sharding": "block i -> rank i mod N, no collective",
                             "compression_ratio": round(n * B / comp_bytes, 3),
                             "bit_exact": "encode == oracle on %d sampled blocks; decode == original on all %d blocks" % (len(sample_idx), n),
                             "setup_encode_GBps": round(n * B / (enc_ms * 1e-3) / 1e9, 2),
                             "cache_flush": "512 MiB memset before every timed step; value = rate of the timed steps (HIP events)" if flush else "none (buffers >= 2 GiB)"}
            traffic, traffic_src = lookup_traffic(mname, n, B, a.dist, param)
            out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                               "kernel": ("k_lz4_index + k_lz4_dec_seq (one decode call)" if is_lz4
                                          else "k_zplan+k_zhufw+k_zchain+k_zmat+k_zexec (one decode call)"),
                               "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": algo_bytes}
            if want_cpu:
                out["cpu_baseline"] = cpu_baseline(ora, np, 0 if is_lz4 else 1, False, param, cpu_comps, [len(c) for c in cpu_comps], B)
            print(json.dumps(out), flush=True)

    # =============================== compress + decompress (configs[2]) ===============================
    elif a.workload in ("lz4", "zstd"):
        method, param, mname = (METHOD_ZSTD, a.level, "zstd") if a.workload == "zstd" else (METHOD_LZ4, a.accel, "lz4")
        stride = (bound(method, B) + 15) & ~15
        d_comp, d_sizes, d_off = alloc(n * stride), alloc(4 * n), alloc(8 * n)
        d_off.upload(np.arange(n, dtype=np.uint64) * np.uint64(stride))
        enc_ms, dec_ms, dec2_ms = [], [], []
        gap_ms = float(os.environ.get("CRYO_BENCH_GAP_MS", "0"))

        def step(timed):
            codec.timer_start()
            codec.compress_batch(method, param, d_raw, B, B, n, d_comp, stride, d_sizes, d_status)
            t1 = codec.timer_stop()
            if gap_ms:   # diagnostic (CRYO_BENCH_GAP_MS): idle time between the passes, outside both timed regions
                codec.sync()
                time.sleep(gap_ms * 1e-3)
            codec.timer_start()
            codec.decompress_batch(method, d_comp, d_off, d_sizes, d_out, B, B, n, d_status)
            t2 = codec.timer_stop()
            if os.environ.get("CRYO_BENCH_DEC_AGAIN"):   # diagnostic: the same decode once more, right behind the first
                codec.timer_start()
                codec.decompress_batch(method, d_comp, d_off, d_sizes, d_out, B, B, n, d_status)
                dec2_ms.append(codec.timer_stop())
            if timed:
                enc_ms.append(t1)
                dec_ms.append(t2)
        steps = min(a.steps, 20) if a.steps == 200 else a.steps   # an encode pass is 10-50x a decode pass
        for _ in range(min(a.warmup, 2)):
            step(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        codec.sync()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        verify_all()
        if rank == 0 and os.environ.get("CRYO_BENCH_TRACE"):
            print("[bench trace] encode ms: %s" % " ".join("%.2f" % x for x in enc_ms), file=sys.stderr)
            print("[bench trace] decode ms: %s" % " ".join("%.2f" % x for x in dec_ms), file=sys.stderr)
            if dec2_ms:
                print("[bench trace] decode again ms: %s" % " ".join("%.2f" % x for x in dec2_ms), file=sys.stderr)
        sizes = d_sizes.download(dtype=np.uint32)
        for i in sorted(set(list(range(0, n, max(1, n // 8)))[:8] + [n - 1])):
            c = d_comp.download(int(sizes[i]), offset=i * stride)
            assert np.array_equal(c, oracle_encode(method, param, i)), "device encode differs from oracle at block %d" % i
        if rank == 0:
            comp_bytes = int(sizes.astype(np.uint64).sum())
            e, d = float(np.mean(enc_ms)), float(np.mean(dec_ms))
            algo = 2 * (n * B + comp_bytes)
            out = dict(base, steps=steps, metric="%s_compress_plus_decompress_uncompressed_GBps" % a.workload,
                       value=round(2 * n * world * B * steps / elapsed / 1e9, 2), unit="GB/s",
                       ms_per_step=round(elapsed / steps * 1e3, 3))
            out.update(per_rank_fields(steps, 2 * n * B))
            out["config"] = {"workload": "%s param %d: compress + decompress %d x %d KiB synthetic cryo blocks per GPU (BASELINE configs[2] shape)"
                             % (a.workload, param, n, B // 1024), "distribution": a.dist, "blocks_per_gpu": n,
                             "compression_ratio": round(n * B / comp_bytes, 3),
                             "encode_GBps": round(n * B / (e * 1e-3) / 1e9, 2), "decode_GBps": round(n * B / (d * 1e-3) / 1e9, 2),
                             "bit_exact": "encode == oracle (libzstd 1.4.8 / liblz4 1.9.3 pinned) on sampled blocks; decode == original on all blocks"}
            out["roofline"] = {"bound": "hbm", "achieved": round(algo / ((e + d) * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": round(algo / ((e + d) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                               "traffic": lookup_traffic(mname + "_roundtrip", n, B, a.dist, param)[0],
                               "traffic_source": lookup_traffic(mname + "_roundtrip", n, B, a.dist, param)[1], "kernel": "encode + decode pair",
                               "encode_frac": round((n * B + comp_bytes) / (e * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                               "decode_frac": round((n * B + comp_bytes) / (d * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                               "algorithmic_bytes_per_launch": algo}
            if want_cpu:
                m = 0 if method == METHOD_LZ4 else 1
                comps = [d_comp.download(int(sizes[i]), offset=i * stride) for i in cpu_idx]
                raws = [d_raw.download(B, offset=i * B) for i in cpu_idx]
                dec = cpu_baseline(ora, np, m, False, param, comps, [len(c) for c in comps], B)
                enc = cpu_baseline(ora, np, m, True, param, raws, [B] * len(raws), B, reps=3)
                # flat keys: the pair rate of the host doing both passes, 1 thread and all threads
                pair = lambda x, y: round(2.0 / (1.0 / x + 1.0 / y), 3)
                cb = {"unit": "GB/s", "cores": 1, "kind": dec["kind"], "sample": dec["sample"], "library": dec.get("library"),
                      "value": pair(enc["value"], dec["value"]), "decode": dec, "encode": enc}
                if "all_cores_value" in dec and "all_cores_value" in enc:
                    cb["all_cores_value"] = pair(enc["all_cores_value"], dec["all_cores_value"])
                    cb["threads"] = dec["threads"]
                out["cpu_baseline"] = cb
            print(json.dumps(out), flush=True)

    # =============================== mixed batch (configs[4]) ===============================
    else:
        zl, acc = 22, 50
        stride = (max(bound(METHOD_LZ4, B), bound(METHOD_ZSTD, B)) + 15) & ~15
        ne, no = (n + 1) // 2, n // 2                     # even job slots: zstd level 22; odd: lz4 acceleration 50
        # lz4 half: GPU encoder over the odd slots (strided view of d_raw)
        d_lz, d_lzs, d_lzo, d_lzst = alloc(no * stride), alloc(4 * no), alloc(8 * no), alloc(4 * no)
        # zstd half: the GPU encoder at level 22 (btultra2, zstd_opt.h) over the even slots; seconds, untimed setup
        d_zs, d_zss, d_zso, d_zsst = alloc(ne * stride), alloc(4 * ne), alloc(8 * ne), alloc(4 * ne)
        codec.timer_start()
        codec._chk(codec.L.cryo_codec_compress_batch(codec.h, METHOD_ZSTD, zl, d_raw.ptr, 2 * B, B, ne, d_zs.ptr, stride, d_zss.ptr, d_zsst.ptr), "compress_batch")
        z_enc_ms = codec.timer_stop()
        assert (d_zsst.download(dtype=np.int32) == 0).all(), "zstd level 22 encode status"
        zsz = d_zss.download(dtype=np.uint32)
        for k in (0, ne // 2, ne - 1):
            c = d_zs.download(int(zsz[k]), offset=k * stride)
            assert np.array_equal(c, ora.zstd_compress(ora.synth(0, job_block(2 * k), B, dist_id), zl)), "zstd level 22 encode differs from oracle"
        d_zso.upload(np.arange(ne, dtype=np.uint64) * np.uint64(stride))
        d_lzo.upload(np.arange(no, dtype=np.uint64) * np.uint64(stride))
        # odd raw blocks start at d_raw + B with stride 2B
        codec.timer_start()
        codec._chk(codec.L.cryo_codec_compress_batch(codec.h, METHOD_LZ4, acc, d_raw.ptr + B, 2 * B, B, no, d_lz.ptr, stride, d_lzs.ptr, d_lzst.ptr), "compress_batch")
        lz_enc_ms = codec.timer_stop()
        assert (d_lzst.download(dtype=np.int32) == 0).all(), "lz4 encode status"
        lzsz = d_lzs.download(dtype=np.uint32)
        for k in (0, no // 2, no - 1):
            c = d_lz.download(int(lzsz[k]), offset=k * stride)
            assert np.array_equal(c, ora.lz4_compress(ora.synth(0, job_block(2 * k + 1), B, dist_id), acc)), "lz4 accel 50 encode differs from oracle"

        def step():
            codec._chk(codec.L.cryo_codec_decompress_batch(codec.h, METHOD_ZSTD, d_zs.ptr, d_zso.ptr, d_zss.ptr, d_out.ptr, 2 * B, B, ne, d_zsst.ptr), "decompress_batch")
            codec._chk(codec.L.cryo_codec_decompress_batch(codec.h, METHOD_LZ4, d_lz.ptr, d_lzo.ptr, d_lzs.ptr, d_out.ptr + B, 2 * B, B, no, d_lzst.ptr), "decompress_batch")
        steps = min(a.steps, 50) if a.steps == 200 else a.steps
        for _ in range(min(a.warmup, 2)):
            step()
        codec.sync()
        barrier()
        t0 = time.perf_counter()
        ms = []
        for _ in range(steps):
            codec.timer_start()
            step()
            ms.append(codec.timer_stop())
        codec.sync()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        assert (d_zsst.download(dtype=np.int32) == 0).all() and (d_lzst.download(dtype=np.int32) == 0).all(), "decode status"
        d_status.memset(0)
        verify_all()
        if rank == 0:
            zbytes, lbytes = int(zsz.astype(np.uint64).sum()), int(lzsz.astype(np.uint64).sum())
            algo = zbytes + lbytes + n * B
            avg_ms = float(np.mean(ms))
            out = dict(base, steps=steps, metric="mixed_zstd22_lz4a50_decompress_uncompressed_GBps",
                       value=round(n * world * B * steps / elapsed / 1e9, 2), unit="GB/s", ms_per_step=round(elapsed / steps * 1e3, 3))
            out.update(per_rank_fields(steps, n * B))
            out["config"] = {"workload": "BASELINE configs[4]: mixed batch of %d x %d KiB blocks per GPU, even = zstd level 22, odd = lz4 acceleration 50; decode of both"
                             % (n, B // 1024), "distribution": a.dist, "blocks_per_gpu": n,
                             "ratio_zstd22": round(ne * B / zbytes, 3), "ratio_lz4_a50": round(no * B / lbytes, 3),
                             "ratio_batch": round(n * B / (zbytes + lbytes), 3),
                             "zstd22_encode_GBps": round(ne * B / (z_enc_ms * 1e-3) / 1e9, 3),
                             "lz4_a50_encode_GBps": round(no * B / (lz_enc_ms * 1e-3) / 1e9, 2),
                             "bit_exact": "lz4 and zstd encode == oracle on sampled blocks; decode of both halves == original on all blocks"}
            out["roofline"] = {"bound": "hbm", "achieved": round(algo / (avg_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": round(algo / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                               "kernel": "zstd decode pipeline + lz4 decode (one step)", "avg_launch_ms": round(avg_ms, 4),
                               "algorithmic_bytes_per_launch": algo}
            if want_cpu:   # the stock libraries over a sample of each half: the rate of a host decoding such a mixed batch
                kz = list(range(0, ne, max(1, ne // max(1, ncpu // 2))))[:max(1, ncpu // 2)]
                kl = list(range(0, no, max(1, no // max(1, ncpu // 2))))[:max(1, ncpu // 2)]
                zc = [d_zs.download(int(zsz[k]), offset=k * stride) for k in kz]
                lc = [d_lz.download(int(lzsz[k]), offset=k * stride) for k in kl]
                dz = cpu_baseline(ora, np, 1, False, zl, zc, [len(c) for c in zc], B)
                dl = cpu_baseline(ora, np, 0, False, acc, lc, [len(c) for c in lc], B)
                pair = lambda x, y: round(2.0 / (1.0 / x + 1.0 / y), 3)   # equal uncompressed bytes in both halves
                cb = {"unit": "GB/s", "cores": 1, "kind": dz["kind"], "value": pair(dz["value"], dl["value"]),
                      "sample": "%d zstd-22 + %d lz4-50 streams of the batch, each half decoded by its stock library; harmonic mean (equal bytes)" % (len(zc), len(lc)),
                      "zstd": dz, "lz4": dl}
                if "all_cores_value" in dz and "all_cores_value" in dl:
                    cb["all_cores_value"] = pair(dz["all_cores_value"], dl["all_cores_value"])
                    cb["threads"] = dz["threads"]
                out["cpu_baseline"] = cb
            print(json.dumps(out), flush=True)

    for b in bufs:
        b.free()
    codec.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
