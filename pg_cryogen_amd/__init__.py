"""pg_cryogen_amd -- MI355X-native cryo-block codec (the hot path of adjust/pg_cryogen).

Layout (only what the path needs):
  csrc/      gfx950 HIP kernels + the C-ABI host code  -> libcryo_codec.so
  host/      C mirror of the reference's compression.h over the C ABI, page-chain write/read
             staging, the decompressed-block cache and the scan iterator  -> libcryo_host.so
  codec.py   ctypes binding used by tests/ and bench.py
The public contract is include/cryo_codec.h.
"""
from .codec import (Codec, CryoError, METHOD_LZ4, METHOD_ZSTD, bound, checksum64,  # noqa: F401
                    device_count, version, DIST_NAMES)
