"""GPU tests of the N > 1 path on a 1-GPU box: two ranks / two handles sharing device 0."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_share_one_gpu():
    """`python bench.py --gpus 2` with no launcher: the parent starts both ranks before touching the GPU, each rank
    decodes its own shard (block i of the job on rank i mod 2), every decoded block is verified on the device, and
    rank 0 reports n_gpus = 2 (BASELINE configs[3] shape at a small size)."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--blocks", "1024", "--cpu-blocks", "128"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["config"]["total_blocks"] == 2048 and j["value"] > 0
    assert "decode == original on all 1024 blocks" in j["config"]["bit_exact"]
    # the measurement contract's two objects are there at N > 1 too
    r, c = j["roofline"], j["cpu_baseline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and "traffic" in r
    assert c["unit"] == "GB/s" and c["value"] > 0 and c["cores"] == 1 and c["kind"] in ("reference", "port") and c["sample"]
    # every rank's own time is on the line: one aggregate would hide a straggler GPU
    assert len(j["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in j["per_rank_ms_per_step"])
    assert j["slowest_rank"] in (0, 1) and j["ms_per_step"] == pytest.approx(max(j["per_rank_ms_per_step"]), rel=1e-3)
    assert j["slowest_rank_GBps"] > 0


def test_bench_eight_ranks_share_one_gpu():
    """the driver's 8-GPU shape through self_launch on this box: eight processes, one rendezvous, MAX-reduce, one line
    (8 x 1 024 blocks; --ref-gbps stands in for the one-GPU figure of the same share)"""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                          "--blocks", "1024", "--cpu-blocks", "64", "--ref-gbps", "100"], capture_output=True, text=True, timeout=900,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["total_blocks"] == 8192 and j["value"] > 0
    assert len(j["per_rank_ms_per_step"]) == 8 and 0 <= j["slowest_rank"] < 8
    assert j["per_gpu_efficiency_vs_n1"] == pytest.approx(j["slowest_rank_GBps"] / 100.0, rel=1e-2) and j["n1_reference_GBps"] == 100


def test_bench_two_ranks_full_per_gpu_share():
    """BASELINE configs[3] at its real per-GPU share: two ranks of 131 072 blocks (16 GiB of output each) sharing this GPU --
    block i of the job on rank i mod 2, every decoded block compared on the device, rank 0 prints roofline and cpu_baseline
    (VERDICT r04 weak #1b: the 8-GPU configs had only run at 1 024 blocks per rank)."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--cpu-blocks", "128"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["blocks_per_gpu"] == 131072 and j["config"]["total_blocks"] == 262144
    assert "decode == original on all 131072 blocks" in j["config"]["bit_exact"]
    assert j["roofline"]["traffic"] is not None, "the 131 072-block shape has a committed counter pass (profiles/r05n_*)"
    assert j["cpu_baseline"]["value"] > 0


def test_two_handles_one_process(oracle):
    """two codec handles in one process (the product dispatcher opens one per device): interleaved calls, each with
    its own stream, workspace and staging buffers"""
    from pg_cryogen_amd import Codec, METHOD_LZ4, METHOD_ZSTD
    B = 131072
    blocks = [oracle.synth(4, i, B, i % 5) for i in range(12)]
    with Codec(0) as a, Codec(0) as b:
        ca = a.compress_blocks(METHOD_LZ4, 1, blocks[:6])
        cb = b.compress_blocks(METHOD_ZSTD, 1, blocks[6:])
        oa, sa = b.decompress_blocks(METHOD_LZ4, ca, B)     # decoded by the OTHER handle
        ob, sb = a.decompress_blocks(METHOD_ZSTD, cb, B)
        assert (sa == 0).all() and (sb == 0).all()
        for x, y in zip(blocks[:6], oa):
            assert np.array_equal(x, y)
        for x, y in zip(blocks[6:], ob):
            assert np.array_equal(x, y)
        for x, c in zip(blocks[:6], ca):
            assert np.array_equal(c, oracle.lz4_compress(x, 1))


def test_multi_handle_dispatcher_round_robin(oracle):
    """cryo_multi_*: block i of a call goes to handle i mod G, one host thread per handle; on a 1-GPU
    box the G handles share device 0.  Bytes and statuses are those of the single-handle calls."""
    import ctypes as C
    from pg_cryogen_amd import codec as cc, METHOD_LZ4, METHOD_ZSTD
    L = cc.lib()
    B, n = 131072, 23
    blocks = [oracle.synth(6, i, B, i % 5) for i in range(n)]
    raw = np.concatenate(blocks)
    for G in (2, 3):
        h = C.c_void_p()
        devs = (C.c_int * G)(*([0] * G))
        assert L.cryo_multi_open(devs, G, C.byref(h)) == 0 and L.cryo_multi_count(h) == G
        for method, param in ((METHOD_LZ4, 1), (METHOD_ZSTD, 1)):
            stride = (cc.bound(method, B) + 15) & ~15
            out = np.zeros(n * stride, np.uint8)
            sizes = np.zeros(n, np.uint32)
            rc = L.cryo_multi_compress_blocks(h, method, param, raw.ctypes.data, B, n, out.ctypes.data, stride, sizes.ctypes.data)
            assert rc == 0, L.cryo_multi_last_error(h)
            comps = []
            for i, b in enumerate(blocks):
                exp = oracle.lz4_compress(b, param) if method == METHOD_LZ4 else oracle.zstd_compress(b, param)
                got = out[i * stride:i * stride + int(sizes[i])]
                assert np.array_equal(got, exp), (G, method, i)
                comps.append(np.ascontiguousarray(got))
            comps[5] = comps[5][:100].copy()                     # one truncated stream among them
            ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in comps])
            csz = np.array([len(c) for c in comps], np.uint32)
            dec = np.zeros(n * B, np.uint8)
            st = np.zeros(n, np.int32)
            rc = L.cryo_multi_decompress_blocks(h, method, ptrs, csz.ctypes.data, n, dec.ctypes.data, B, st.ctypes.data)
            assert rc == 0, L.cryo_multi_last_error(h)
            for i, b in enumerate(blocks):
                if i == 5:
                    assert st[i] != 0
                else:
                    assert st[i] == 0 and np.array_equal(dec[i * B:(i + 1) * B], b), (G, method, i)
        L.cryo_multi_close(h)
