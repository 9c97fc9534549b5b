"""Every BASELINE.json config through bench.py itself, at sizes that take seconds (the driver's GPU run sees them):
configs[1] lz4_decode, configs[2] zstd (compress + decompress; and lz4), zstd_decode, configs[4] mixed.  Each run
verifies its bytes inside bench.py (encode == oracle on sampled blocks, decode == original on every block) and must
print the roofline and cpu_baseline objects of the measurement contract."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CRYO_CODEC_LIB"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                         timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("workload,metric,blocks", [
    ("lz4_decode", "lz4_decompress_uncompressed_GBps", 2048),
    ("zstd_decode", "zstd_decompress_uncompressed_GBps", 2048),
    ("zstd", "zstd_compress_plus_decompress_uncompressed_GBps", 2048),      # BASELINE configs[2]
    ("lz4", "lz4_compress_plus_decompress_uncompressed_GBps", 2048),
    ("mixed", "mixed_zstd22_lz4a50_decompress_uncompressed_GBps", 1024),    # BASELINE configs[4]: level-22 encode on the GPU
])
def test_bench_workload_line(workload, metric, blocks):
    j = run_bench("--workload", workload, "--blocks", str(blocks), "--steps", "2", "--warmup", "1", "--cpu-blocks", "128")
    assert j["metric"] == metric and j["unit"] == "GB/s" and j["value"] > 0 and j["n_gpus"] == 1
    assert j["dtype"] == "u8" and j["data"] == "synthetic" and j["vs_baseline"] is None and j["scaling"] == "weak"
    cfg = j["config"]
    assert "workload" in cfg and "model" not in cfg
    be = cfg["bit_exact"]
    assert "== oracle" in be and ("decode == original on all" in be or "== original on all blocks" in be), be
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert 0 < r["achieved"] < r["peak"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = j["cpu_baseline"]
    assert c["unit"] == "GB/s" and c["value"] > 0 and c["cores"] == 1 and c["kind"] in ("reference", "port") and c["sample"]


@pytest.mark.parametrize("workload,blocks", [
    ("zstd", 65536),     # BASELINE configs[2] at its full size: zstd level 1 compress + decompress of the 64 Ki-block batch
    ("mixed", 16384),    # BASELINE configs[4] at the per-GPU size bench.py defaults to (level-22 encode on the GPU: seconds)
])
def test_bench_full_size_configs(workload, blocks):
    """configs[2] and configs[4] at FULL size through bench.py, so that the driver's GPU run -- not only the builder's --
    has seen them.  bench.py verifies the bytes itself: encode == oracle on sampled blocks, every
    decoded block == its original on the device."""
    j = run_bench("--workload", workload, "--blocks", str(blocks), "--steps", "2", "--warmup", "1", "--cpu-blocks", "128")
    assert j["config"]["blocks_per_gpu"] == blocks and j["value"] > 0
    assert "== oracle" in j["config"]["bit_exact"] and "original on all blocks" in j["config"]["bit_exact"]
    assert 0 < j["roofline"]["frac"] < 1 and j["cpu_baseline"]["value"] > 0


def test_automatic_walkers_batch_against_oracle(codec, oracle):
    """A 16 384-block batch takes the index pass at its AUTOMATIC setting (four walkers per block, lz4_decode_plan) --
    the other tests force the walker count on a few dozen blocks.  Blocks are generated and compressed on the device;
    72 sampled blocks are checked against the oracle (generator, compressed bytes, decoded bytes), all of them against
    their originals on the device."""
    from pg_cryogen_amd import METHOD_LZ4, bound
    B, n = 131072, 16384
    stride = (bound(METHOD_LZ4, B) + 15) & ~15
    d_raw, d_comp, d_out = codec.alloc(n * B), codec.alloc(n * stride), codec.alloc(n * B)
    d_sizes, d_off, d_st, d_mis = codec.alloc(4 * n), codec.alloc(8 * n), codec.alloc(4 * n), codec.alloc(8)
    try:
        codec.synth_batch(9, 0, n, B, 0, d_raw)
        codec.compress_batch(METHOD_LZ4, 1, d_raw, B, B, n, d_comp, stride, d_sizes, d_st)
        assert (d_st.download(dtype=np.int32) == 0).all()
        sizes = d_sizes.download(dtype=np.uint32)
        d_off.upload(np.arange(n, dtype=np.uint64) * np.uint64(stride))
        d_out.memset(0xEE)
        codec.decompress_batch(METHOD_LZ4, d_comp, d_off, d_sizes, d_out, B, B, n, d_st)
        codec.sync()
        assert (d_st.download(dtype=np.int32) == 0).all()
        d_mis.memset(0)
        codec.compare_batch(d_raw, B, d_out, B, B, n, d_mis)
        codec.sync()
        assert int(d_mis.download(dtype=np.uint64)[0]) == 0
        for i in sorted(set(list(range(0, n, n // 64)) + [1, 2, 3, n - 3, n - 2, n - 1, 4097, 8191])):
            raw = oracle.synth(9, i, B, 0)
            comp = d_comp.download(int(sizes[i]), offset=i * stride)
            assert np.array_equal(comp, oracle.lz4_compress(raw, 1)), i
            r, exp = oracle.lz4_decompress(comp, B)
            assert r == B and np.array_equal(d_out.download(B, offset=i * B), exp), i
    finally:
        for x in (d_raw, d_comp, d_out, d_sizes, d_off, d_st, d_mis):
            x.free()
