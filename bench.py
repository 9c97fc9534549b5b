#!/usr/bin/env python3
"""bench.py -- headline benchmark of the cryo-block codec hot path on MI355X.

Metric (BASELINE.json): uncompressed GB/s of LZ4 decompress over a batch of
synthetic 128 KiB cryo blocks resident in HBM (configs[1]: 64k x 128 KiB on one
GPU; weak scaling: every rank decodes its own 64k blocks, block i of the job
belongs to rank i mod N, no collective on the data path).

A "step" = one decode pass over the rank's whole batch (one kernel launch).
Setup (untimed): blocks are generated on the device, compressed on the device
by the bit-exact HIP LZ4 encoder, and a strided sample is verified against the
CPU oracle; after the timed region every decoded block is compared on the
device with its original.

Prints ONE JSON line on rank 0 (see the driver contract in the task).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=65536, help="cryo blocks per GPU")
    ap.add_argument("--block-size", type=int, default=131072)
    ap.add_argument("--dist", default="wide", choices=["wide", "narrow", "int4", "random", "zeros"])
    ap.add_argument("--accel", type=int, default=1, help="lz4 acceleration used to produce the inputs")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="diagnostic ablation runs only: result is not valid")
    ap.add_argument("--workload", default="lz4_decode", choices=["lz4_decode", "zstd_decode", "zstd", "lz4"],
                    help="lz4_decode = headline (BASELINE configs[1]); zstd_decode = same shape for the zstd method; zstd / lz4 = compress+decompress of the same "
                         "batch (configs[2] shape), a secondary measurement")
    ap.add_argument("--level", type=int, default=1, help="zstd level for --workload zstd")
    return ap.parse_args()


def cpu_baseline(comps, B, budget_s, method=0):
    """cpu_baseline leg: the oracle ("port", 1 thread) and -- reported beside it -- the stock library the
    reference links (reference compression.c:84/116 call shape), 1 thread and all cores, timed by
    oracle/cpu_bench.c on this machine's host cores over the same sample of compressed blocks."""
    import oracle_lib
    ora = oracle_lib.Oracle()
    fn = ora.L.cryo_oracle_cpu_decode_bench
    fn.restype = ctypes.c_double
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                   ctypes.c_uint32, ctypes.c_int, ctypes.c_double, ctypes.c_char_p, ctypes.c_size_t]
    offs = np.zeros(len(comps), np.uint64)
    pos = 0
    for i, c in enumerate(comps):
        offs[i] = pos
        pos += (len(c) + 63) & ~63
    packed = np.zeros(pos + 64, np.uint8)
    for i, c in enumerate(comps):
        packed[int(offs[i]):int(offs[i]) + len(c)] = c
    sizes = np.array([len(c) for c in comps], np.uint32)
    ver = ctypes.create_string_buffer(64)

    def run(stock, threads, secs):
        return fn(method, stock, packed.ctypes.data, offs.ctypes.data, sizes.ctypes.data, len(comps), B, threads,
                  secs, ver, 64)
    name = "lz4" if method == 0 else "zstd"
    v = run(0, 1, budget_s * 0.4)
    res = {"value": round(v, 3), "unit": "GB/s", "cores": 1, "kind": "port",
           "sample": "%d distinct compressed blocks (%.1f MiB uncompressed) decoded repeatedly for %.1f s by "
                     "oracle/%s, 1 thread" % (len(comps), len(comps) * B / 2**20, budget_s * 0.4,
                                              "lz4_oracle.c" if method == 0 else "zstd_dec_oracle.c")}
    T = os.cpu_count() or 1
    one = run(1, 1, budget_s * 0.25)
    if one > 0:
        allc = run(1, T, budget_s * 0.35)
        res["stock_lib%s" % name] = {"version": ver.value.decode(),
                                     "call": "LZ4_decompress_safe(src,dst,csize,B)" if method == 0
                                     else "ZSTD_decompress(dst,B,src,csize)",
                                     "GBps_1_thread": round(one, 3), "GBps_all_threads": round(allc, 3), "threads": T}
    return res


def bench_roundtrip(a, codec, rank, world, barrier, torch, dist):
    """secondary workload: compress + decompress of the rank's batch (BASELINE configs[2] shape)"""
    from pg_cryogen_amd import METHOD_LZ4, METHOD_ZSTD, bound
    from pg_cryogen_amd.codec import DIST_NAMES
    import oracle_lib
    method = METHOD_ZSTD if a.workload == "zstd" else METHOD_LZ4
    param = a.level if method == METHOD_ZSTD else a.accel
    B, n = a.block_size, a.blocks
    dist_id = DIST_NAMES.index(a.dist)
    stride = (bound(method, B) + 15) & ~15
    d_raw, d_comp, d_out = codec.alloc(n * B), codec.alloc(n * stride), codec.alloc(n * B)
    d_sizes, d_status, d_off, d_mis = codec.alloc(4 * n), codec.alloc(4 * n), codec.alloc(8 * n), codec.alloc(8)
    codec.synth_batch(0, rank, n, B, dist_id, d_raw, block_step=world)
    d_off.upload(np.arange(n, dtype=np.uint64) * np.uint64(stride))
    enc_ms, dec_ms = [], []

    def step(timed):
        codec.timer_start()
        codec.compress_batch(method, param, d_raw, B, B, n, d_comp, stride, d_sizes, d_status)
        t1 = codec.timer_stop()
        codec.timer_start()
        codec.decompress_batch(method, d_comp, d_off, d_sizes, d_out, B, B, n, d_status)
        t2 = codec.timer_stop()
        if timed:
            enc_ms.append(t1)
            dec_ms.append(t2)
    for _ in range(a.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    codec.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    assert (d_status.download(dtype=np.int32) == 0).all()
    d_mis.memset(0)
    codec.compare_batch(d_raw, B, d_out, B, B, n, d_mis)
    codec.sync()
    assert int(d_mis.download(dtype=np.uint64)[0]) == 0, "round trip mismatch"
    sizes = d_sizes.download(dtype=np.uint32)
    ora = oracle_lib.Oracle()
    for i in sorted(set(list(range(0, n, max(1, n // 8)))[:8] + [n - 1])):
        c = d_comp.download(int(sizes[i]), offset=i * stride)
        raw = ora.synth(0, rank + i * world, B, dist_id)
        exp = ora.zstd_compress(raw, param) if method == METHOD_ZSTD else ora.lz4_compress(raw, param)
        assert np.array_equal(c, exp), "device encode differs from oracle at block %d" % i
    t = torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        comp_bytes = int(sizes.astype(np.uint64).sum())
        e, d = float(np.mean(enc_ms)), float(np.mean(dec_ms))
        algo = 2 * (n * B + comp_bytes)
        print(json.dumps({
            "metric": "%s_compress_plus_decompress_uncompressed_GBps" % a.workload,
            "value": round(2 * n * world * B * a.steps / float(t[0]) / 1e9, 2), "unit": "GB/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(float(t[0]) / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s param %d: compress + decompress %d x %d KiB synthetic cryo blocks per GPU"
                                   % (a.workload, param, n, B // 1024), "distribution": a.dist,
                       "compression_ratio": round(n * B / comp_bytes, 3),
                       "encode_GBps": round(n * B / (e * 1e-3) / 1e9, 2), "decode_GBps": round(n * B / (d * 1e-3) / 1e9, 2),
                       "bit_exact": "encode == oracle (libzstd 1.4.8 / liblz4 1.9.3 pinned) on sampled blocks; "
                                    "decode == original on all blocks"},
            "roofline": {"bound": "hbm", "achieved": round(algo / ((e + d) * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(algo / ((e + d) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "traffic": None, "kernel": "encode+decode pair"}}), flush=True)


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world

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

    from pg_cryogen_amd import Codec, METHOD_LZ4, bound
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
    if not a.workload.endswith("_decode"):
        bench_roundtrip(a, codec, rank, world, barrier, torch, dist)
        codec.close()
        if world > 1:
            dist.destroy_process_group()
        return
    from pg_cryogen_amd import METHOD_ZSTD
    is_lz4 = a.workload == "lz4_decode"
    method, param, mname = (METHOD_LZ4, a.accel, "lz4") if is_lz4 else (METHOD_ZSTD, a.level, "zstd")
    stride = (bound(method, B) + 15) & ~15

    # ---------------- setup (untimed) ----------------
    d_raw = codec.alloc(n * B)
    d_comp = codec.alloc(n * stride)
    d_out = codec.alloc(n * B)
    d_sizes, d_status = codec.alloc(4 * n), codec.alloc(4 * n)
    d_off = codec.alloc(8 * n)
    d_mis = codec.alloc(8)
    # block i of the job lives on rank i mod N: this rank's k-th block is job block rank + k*N
    # (pg_cryogen_amd/shard.py); every block of the job is distinct
    job_block = lambda k: rank + k * world
    codec.synth_batch(0, rank, n, B, dist_id, d_raw, block_step=world)
    codec.timer_start()
    codec.compress_batch(method, param, d_raw, B, B, n, d_comp, stride, d_sizes, d_status)
    enc_ms = codec.timer_stop()
    st = d_status.download(dtype=np.int32)
    assert (st == 0).all(), "encode status"
    sizes = d_sizes.download(dtype=np.uint32)
    d_off.upload(np.arange(n, dtype=np.uint64) * np.uint64(stride))
    comp_bytes = int(sizes.astype(np.uint64).sum())

    # spot check vs the CPU oracle: strided sample of the device-encoded blocks
    ora = oracle_lib.Oracle()
    sample_idx = sorted(set(list(range(0, n, max(1, n // 16)))[:16] + [n - 1]))
    sample_comps = []
    for i in sample_idx:
        c = d_comp.download(int(sizes[i]), offset=i * stride)
        raw = ora.synth(0, job_block(i), B, dist_id)
        exp = ora.lz4_compress(raw, param) if is_lz4 else ora.zstd_compress(raw, param)
        assert np.array_equal(c, exp), "device encode differs from oracle at block %d" % i
        sample_comps.append(c)
    # cpu_baseline sample: 512 device-encoded blocks (64 MiB uncompressed), enough for every host thread
    cpu_idx = list(range(0, n, max(1, n // 512)))[:512]
    cpu_comps = [d_comp.download(int(sizes[i]), offset=i * stride) for i in cpu_idx] if (world == 1 and not a.no_cpu_baseline) else []

    def step():
        codec.decompress_batch(method, d_comp, d_off, d_sizes, d_out, B, B, n, d_status)

    for _ in range(a.warmup):
        step()
    codec.sync()

    # ---------------- timed region ----------------
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(a.steps):
        codec.timer_start()          # HIP events on the codec's own stream
        step()
        kernel_ms.append(codec.timer_stop())
    codec.sync()
    barrier()
    elapsed = time.perf_counter() - t0

    # ---------------- verification (untimed) ----------------
    st = d_status.download(dtype=np.int32)
    assert a.no_verify or (st == 0).all(), "decode status"
    d_mis.memset(0)
    codec.compare_batch(d_raw, B, d_out, B, B, n, d_mis)
    codec.sync()
    mismatch = int(d_mis.download(dtype=np.uint64)[0])
    assert a.no_verify or mismatch == 0, "decoded blocks differ from originals: %d" % mismatch

    t = torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0])

    if rank == 0:
        total_blocks = n * world
        value = total_blocks * B * a.steps / elapsed / 1e9
        avg_ms = float(np.mean(kernel_ms))
        algo_bytes = comp_bytes + n * B        # per launch: compressed bytes read + B written per block
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None   # HBM bytes per launch from the PMC passes of the same workload (profiles/)
        try:
            for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
                if fn.endswith("_hbm_traffic.json"):
                    t = json.load(open(os.path.join(ROOT, "profiles", fn)))
                    wl = t.get("workload", {})
                    if (wl.get("method", "lz4"), wl.get("blocks_per_gpu"), wl.get("block_size"), wl.get("distribution"),
                            wl.get("param", wl.get("lz4_acceleration"))) == (mname, n, B, a.dist, param):
                        traffic = t["traffic_bytes_per_launch"]
                        break
        except OSError:
            pass
        out = {
            "metric": "%s_decompress_uncompressed_GBps" % mname, "value": round(value, 2), "unit": "GB/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s decompress %d x %d KiB synthetic cryo blocks per GPU" % (mname.upper(), n, B // 1024),
                       "distribution": a.dist, "method": mname, "param": param, "block_size": B,
                       "blocks_per_gpu": n, "sharding": "block i -> rank i mod N, no collective",
                       "compression_ratio": round(n * B / comp_bytes, 3),
                       "bit_exact": "encode == oracle on %d sampled blocks; decode == original on all %d blocks"
                                    % (len(sample_idx), n),
                       "setup_encode_GBps": round(n * B / (enc_ms * 1e-3) / 1e9, 2)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": "k_lz4_dec_ring" if is_lz4 else "k_zplan+k_zhuf+k_zseq+k_zexec (one decode call)",
                         "avg_launch_ms": round(avg_ms, 4),
                         "algorithmic_bytes_per_launch": algo_bytes},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cpu_comps, B, a.cpu_seconds, 0 if is_lz4 else 1)
        print(json.dumps(out), flush=True)

    for b in (d_raw, d_comp, d_out, d_sizes, d_status, d_off, d_mis):
        b.free()
    codec.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
