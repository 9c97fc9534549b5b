"""Property tests (hypothesis) of the CPU oracle: decode(encode(x)) == x on arbitrary bytes, and the
decoders never write outside their output on arbitrary (mostly malformed) input."""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle_lib

ORA = oracle_lib.Oracle()


@settings(max_examples=150, deadline=None)
@given(st.binary(min_size=0, max_size=3000), st.integers(0, 50))
def test_lz4_oracle_roundtrip_arbitrary_bytes(data, accel):
    a = np.frombuffer(data, np.uint8)
    c = ORA.lz4_compress(a, accel)
    r, out = ORA.lz4_decompress(c, len(a))
    assert r == len(a) and bytes(out[:len(a)]) == data


@settings(max_examples=150, deadline=None)
@given(st.binary(min_size=1, max_size=400), st.integers(1, 512))
def test_decoders_stay_inside_output_on_garbage(data, cap):
    a = np.frombuffer(data, np.uint8)
    for fn in (ORA.L.cryo_oracle_lz4_decompress, ORA.L.cryo_oracle_zstd_decompress):
        out = np.full(cap + 64, 0xC3, np.uint8)
        r = fn(a.ctypes.data, a.nbytes, out.ctypes.data, cap)
        assert r <= cap
        assert (out[cap:] == 0xC3).all()


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2**32), st.integers(0, 4), st.sampled_from([-5, -1, 1, 2]))
def test_zstd_oracle_roundtrip_synthetic(seed, dist, level):
    raw = ORA.synth(seed, seed % 977, 20000, dist)
    c = ORA.zstd_compress(raw, level)
    r, out = ORA.zstd_decompress(c, 20000)
    assert r == 20000 and np.array_equal(out, raw)


def test_short_differential_hunt_of_the_encoder_oracles():
    """a few seconds of tests/hunt_oracle.py inside the suite (zstd at random levels -5 .. 22, LZ4 at random accelerations, on
    tiny / boundary-size / structured / few-sequence / sparse blocks): the oracle equals the live libraries"""
    import subprocess, sys, os
    import pytest
    import oracle_lib
    st = oracle_lib.StockLibs()
    if st.zstd is None or st.lz4 is None:
        pytest.skip("stock libraries not loadable")
    here = os.path.dirname(os.path.abspath(__file__))
    for mode in ("zstd", "lz4"):
        r = subprocess.run([sys.executable, os.path.join(here, "hunt_oracle.py"), "4", "7", mode], capture_output=True, text=True, cwd="/tmp")
        assert r.returncode == 0 and "hunt ok" in r.stdout, r.stdout + r.stderr
