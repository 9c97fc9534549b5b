"""Build-time checks of generated code that no output comparison can see (CPU only: hipcc cross-compiles)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_index_kernel_assembly_loads_are_not_copied():
    """k_lz4_index (lz4_index.hip) and k_zchain4 (zstd_pipe.hip) load their ring chunks with inline-assembly loads and
    hand-written s_waitcnt vmcnt(N): a register copy inserted by the compiler in front of the wait reads stale data, and a
    slot register handed to something else in front of the drain is overwritten by a load still on its way (k_zchain4
    stored into the wild that way before its drain named the slots; in k_lz4_index either only costs speed -- the decoder
    validates every index entry).  tools/check_index_isa.py inspects the generated assembly of both."""
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)   # tests/run_sanitized.sh preloads the sanitizer runtimes: not into the compiler
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_index_isa.py")], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
