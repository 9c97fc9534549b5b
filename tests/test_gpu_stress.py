"""Short runs of the differential stress harness (tests/stress_gpu.py) inside the GPU suite: structured random
blocks against the stock libraries, mutated streams against the oracle.  The long runs are done by hand."""
import sys

import pytest

import oracle_lib
import stress_gpu

pytestmark = pytest.mark.gpu


def _need_stock():
    s = oracle_lib.StockLibs()
    if s.lz4 is None or s.zstd is None:
        pytest.skip("stock liblz4/libzstd not present")


def test_stress_structured_blocks_vs_stock(monkeypatch):
    _need_stock()
    monkeypatch.setattr(sys, "argv", ["stress_gpu.py", "8", "101"])
    stress_gpu.main()


def test_fuzz_mutated_streams_vs_oracle(monkeypatch):
    _need_stock()
    monkeypatch.setattr(sys, "argv", ["stress_gpu.py", "fuzz", "8", "102"])
    stress_gpu.main()
