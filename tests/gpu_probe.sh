#!/bin/bash
# one-off probe of the GPU box environment (not a test)
set -x
nproc; lscpu | grep -E "Model name|Socket|Core|Thread" ; free -g | head -2
ls -la /usr/lib/x86_64-linux-gnu/liblz4.so.1 /usr/lib/x86_64-linux-gnu/libzstd.so.1 /opt/conda/lib/libzstd.so.1 2>&1
rocminfo | grep -E "gfx|Compute Unit" | head -6
echo "--- rocm runtime"; CRYO_HIP_RUNTIME=rocm python -c "
from pg_cryogen_amd import codec, _loader
print(codec.device_count(), _loader.runtime_path())
c = codec.Codec(0); d = c.alloc(1<<20); c.synth_batch(0,0,8,131072,0,d); c.sync(); print(d.download()[:16]); c.close()"
echo "--- torch runtime"; CRYO_HIP_RUNTIME=torch python -c "
import torch
from pg_cryogen_amd import codec, _loader
print(codec.device_count(), _loader.runtime_path())
c = codec.Codec(0); d = c.alloc(1<<20); c.synth_batch(0,0,8,131072,0,d); c.sync(); print(d.download()[:16]);
t = torch.zeros(16, device='cuda'); torch.cuda.synchronize(); print(t.sum().item()); c.close()"
