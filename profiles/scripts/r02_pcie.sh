#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import torch, time
n=512<<20
h=torch.empty(n,dtype=torch.uint8).pin_memory()
d=torch.empty(n,dtype=torch.uint8,device='cuda')
for name,fn in (("H2D pinned",lambda: d.copy_(h,non_blocking=True)),("D2H pinned",lambda: h.copy_(d,non_blocking=True))):
    fn(); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(name, "%.1f GB/s"%(5*n/(time.perf_counter()-t)/1e9))
p=torch.empty(n,dtype=torch.uint8)
p.fill_(1)
for name,fn in (("H2D pageable",lambda: d.copy_(p)),("D2H pageable",lambda: p.copy_(d))):
    fn(); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    print(name, "%.1f GB/s"%(3*n/(time.perf_counter()-t)/1e9))
# both directions at once on two streams
s1,s2=torch.cuda.Stream(),torch.cuda.Stream()
h2=torch.empty(n,dtype=torch.uint8).pin_memory(); d2=torch.empty(n,dtype=torch.uint8,device='cuda')
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5):
    with torch.cuda.stream(s1): d.copy_(h,non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2,non_blocking=True)
torch.cuda.synchronize()
print("H2D + D2H concurrently: %.1f GB/s each"%(5*n/(time.perf_counter()-t)/1e9))
import numpy as np
a=np.ones(n,dtype=np.uint8); b=np.empty(n,dtype=np.uint8); b[:]=0
t=time.perf_counter(); b[:]=a; print("host memcpy 1 thread: %.1f GB/s"%(n/(time.perf_counter()-t)/1e9))
PY
