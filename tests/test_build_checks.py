"""Build-time checks of generated code that no output comparison can see (CPU only: hipcc cross-compiles)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_index_kernel_assembly_loads_are_not_copied():
    """k_lz4_index loads its ring chunks with inline-assembly loads and hand-written s_waitcnt vmcnt(N) (lz4_index.hip):
    a register copy inserted by the compiler in front of the wait would read stale data, and -- the decoder validates
    every index entry -- only cost speed.  tools/check_index_isa.py inspects the generated assembly."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_index_isa.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
