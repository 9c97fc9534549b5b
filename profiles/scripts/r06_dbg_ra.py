import sys, struct, ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pg_cryogen_amd import host
from test_host_plumbing import _load
host.use(production=True)
L = host.lib()
host.set_block_size(1<<20)
L.cryo_define_compression_gucs(); L.cryo_init_cache()
rows = [struct.pack("<i", i) for i in range(1, 10001)]
mem, rel, blocks, firsts = _load(L, rows, 1, host.COMP_LZ4, batch=16)
guc = C.c_int.in_dll(L, "cryo_gpu_readahead_blocks_guc")
for k in (8,1,5,8):
    L.cryo_cache_configure(16); guc.value=k
    it = L.cryo_seqscan_iter_create(); got=[]; seq=[]
    while True:
        b = L.cryo_seqscan_iter_next(it)
        if L.cryo_memrel_nblocks(mem) <= b: break
        e = C.c_int(-1)
        err = L.cryo_read_data(C.byref(rel), it, b, C.byref(e))
        seq.append((b,err,e.value))
        if err == host.CRYO_ERR_EMPTY_BLOCK: continue
        assert err == 0, (b, err)
        got.append(bytes(np.ctypeslib.as_array(C.cast(L.cryo_cache_get_data(e.value), C.POINTER(C.c_uint8)), (1 << 20,))))
    bad=[i for i,(g,bk) in enumerate(zip(got,blocks)) if g!=bk]
    print(k, len(got), bad, seq[:12])
    for i in bad[:3]:
        g=np.frombuffer(got[i],np.uint8); bk=np.frombuffer(blocks[i],np.uint8); d=np.nonzero(g!=bk)[0]
        which=[j for j,x in enumerate(blocks) if x==got[i]]
        print("  block",i,"first diff",d[0],"ndiff",len(d),"equals block",which)
