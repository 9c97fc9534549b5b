"""CPU tests: the C-ABI library loads, exports every symbol include/cryo_codec.h declares,
and fails loudly (no CPU fallback) when no GPU is present."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "cryo_codec.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cryo_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported():
    from pg_cryogen_amd import codec
    L = codec.lib()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libcryo_codec.so does not export %s" % n
    assert sorted(codec.ABI_SYMBOLS) == names


def test_version_and_bounds_without_gpu():
    from pg_cryogen_amd import codec
    assert "gfx950" in codec.version()
    assert codec.bound(codec.METHOD_LZ4, 0) == 16
    assert codec.bound(7, 100) == 0


def test_no_cpu_fallback_without_gpu():
    from pg_cryogen_amd import codec
    if codec.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(codec.CryoError) as e:
        codec.Codec(0)
    assert e.value.code == codec.E_NODEV


def test_product_never_touches_oracle():
    """the product tree must not reference oracle/ (tests, smoke and bench's cpu_baseline may)"""
    bad = []
    for base in ("pg_cryogen_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".so", ".o", ".pyc")):
                    continue
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                if re.search(r"#include\s*[\"<][^\">]*oracle|libcryo_oracle|cryo_oracle_|import\s+oracle|oracle_lib", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_production_host_library_has_no_test_hook():
    """the codec-double hook of the CPU plumbing tests exists only in libcryo_host_test.so"""
    import ctypes
    from pg_cryogen_amd import _loader, host
    _loader.load()
    prod = ctypes.CDLL(host.HOST_LIB_PATH)
    assert not hasattr(prod, "cryo_host_set_codec_ops")
    assert hasattr(prod, "cryo_compress") and hasattr(prod, "cryo_decompress") and hasattr(prod, "cryo_scan_next_batch")
    assert hasattr(ctypes.CDLL(host.HOST_TEST_LIB_PATH), "cryo_host_set_codec_ops")


def test_option_numbers_match_the_header():
    """the Python mirror's OPT_* constants are the header's cryo_option values (a test that flips an option by number
    must flip the one it names)"""
    import re
    from pg_cryogen_amd import codec as cc
    text = open(os.path.join(ROOT, "include", "cryo_codec.h")).read()
    body = text[text.index("typedef enum {", text.index("per-handle options")):text.index("} cryo_option;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    enum = {m.group(1): int(m.group(2)) for m in re.finditer(r"CRYO_OPT_(\w+)\s*=\s*(\d+)", body)}
    assert len(enum) >= 9 and len(set(enum.values())) == len(enum)
    for name, value in enum.items():
        assert getattr(cc, "OPT_" + name) == value, name
