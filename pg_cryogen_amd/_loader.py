"""Locate and load libcryo_codec.so (the gfx950 C-ABI library) with ctypes.

libcryo_codec.so carries no DT_NEEDED on libamdhip64, so that a process ends up
with exactly ONE HIP runtime: if PyTorch is already imported its bundled
``torch/lib/libamdhip64.so`` is reused (device pointers and
``torch.cuda.synchronize()`` then refer to the same runtime), otherwise the
system ROCm runtime is loaded.  ``CRYO_HIP_RUNTIME`` = ``torch`` | ``rocm`` |
an explicit path overrides the choice.

There is no CPU fallback: a missing library is an ImportError with the build
command in the message, and a missing GPU surfaces as CRYO_E_NODEV from
``cryo_codec_open``.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRYO_CODEC_LIB") or os.path.join(_HERE, "libcryo_codec.so")   # override: A/B builds of experiments
_ROCM_RT = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libamdhip64.so")

_lib = None
_runtime_path = None


def _torch_runtime():
    try:
        import torch  # noqa: F401
    except Exception:
        return None
    p = os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib", "libamdhip64.so")
    return p if os.path.exists(p) else None


def _pick_runtime():
    want = os.environ.get("CRYO_HIP_RUNTIME", "")
    if want == "torch":
        p = _torch_runtime()
        if p is None:
            raise ImportError("CRYO_HIP_RUNTIME=torch but torch/lib/libamdhip64.so was not found")
        return p
    if want == "rocm" or want == "":
        if want == "" and "torch" in sys.modules:
            p = _torch_runtime()
            if p is not None:
                return p
        return _ROCM_RT
    return want


def load():
    """Return the ctypes handle of libcryo_codec.so (loaded once)."""
    global _lib, _runtime_path
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pg_cryogen_amd: %s is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C pg_cryogen_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    _runtime_path = _pick_runtime()
    try:
        ctypes.CDLL(_runtime_path, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:
        raise ImportError("pg_cryogen_amd: cannot load the HIP runtime %s: %s" % (_runtime_path, e))
    _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def runtime_path():
    return _runtime_path
