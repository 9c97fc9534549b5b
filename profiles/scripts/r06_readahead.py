"""Round 6: what the transparent read-ahead of cryo_read_data_rel buys an UNCHANGED table AM.
The reference's scan loop -- one iterator pop, one cryo_read_data per block -- over tables of 1 MiB cryo blocks, through the
production libcryo_host.so, against one host thread decoding the same chains with the stock library.
Run on the GPU box: python profiles/scripts/r06_readahead.py > gpurun_out/r06_readahead.txt"""
import ctypes as C, os, struct, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pg_cryogen_amd import host
import oracle_lib
from test_host_plumbing import pack_rows

host.use(production=True)
L = host.lib()
host.set_block_size(1 << 20)
L.cryo_define_compression_gucs()
stock = oracle_lib.StockLibs()
rng = np.random.default_rng(6)


def table(method, nblocks, relid):
    mem = L.cryo_memrel_create(); rel = host.CryoRel(); L.cryo_memrel_bind(mem, relid, C.byref(rel))
    # JSON-ish text rows: ~2.3x on lz4, a few hundred rows per block
    rows = []
    for i in range(nblocks * 2300):
        rows.append(struct.pack("<i", i) + ('{"id": %d, "k": "%s", "v": "%s"}' % (i, "abcdefgh"[i % 8] * 8, rng.bytes(96).hex())).encode())
    blocks = pack_rows(L, rows, 2, 1 << 20)[:nblocks]
    firsts = []
    for i in range(0, len(blocks), 16):
        chunk = blocks[i:i + 16]
        fb = (C.c_uint32 * len(chunk))(*[L.cryo_memrel_reserve(mem) for _ in chunk])
        assert L.cryo_stage_write_batch(C.byref(rel), b"".join(chunk), len(chunk), method, 777, fb) == 0
        firsts += list(fb)
    return mem, rel, blocks, firsts


def scan(mem, rel, k):
    guc = C.c_int.in_dll(L, "cryo_gpu_readahead_blocks_guc"); guc.value = k
    L.cryo_cache_configure(16)
    it = L.cryo_seqscan_iter_create(); n = 0; calls0 = L.cryo_cache_codec_calls()
    t0 = time.perf_counter()
    while True:
        b = L.cryo_seqscan_iter_next(it)
        if L.cryo_memrel_nblocks(mem) <= b: break
        e = C.c_int(-1)
        err = L.cryo_read_data(C.byref(rel), it, b, C.byref(e))
        if err == host.CRYO_ERR_EMPTY_BLOCK: continue
        assert err == 0, err
        n += 1
    dt = time.perf_counter() - t0
    L.cryo_seqscan_iter_free(it)
    return n, dt, L.cryo_cache_codec_calls() - calls0


for mname, method in (("lz4", host.COMP_LZ4), ("zstd", host.COMP_ZSTD)):
    mem, rel, blocks, firsts = table(method, 64, 100 + method)
    # one host thread, stock library, the same blocks
    comps = [(stock.lz4_compress if method == 0 else (lambda r: stock.zstd_compress(r, 1)))(np.frombuffer(b, np.uint8)) for b in blocks]
    t0 = time.perf_counter()
    for c in comps:
        (stock.lz4_decompress if method == 0 else stock.zstd_decompress)(c, 1 << 20)
    host_ms = (time.perf_counter() - t0) / len(comps) * 1e3
    scan(mem, rel, 8)   # warm-up: device open, workspace
    for k in (1, 2, 4, 8):
        best = min(scan(mem, rel, k)[1] for _ in range(3))
        n, _, calls = scan(mem, rel, k)
        print("%-4s 64 x 1 MiB blocks, ratio %.2f: gpu_readahead_blocks = %d: %3d codec calls, %.3f ms per scanned block  (one host thread, stock library: %.3f ms)"
              % (mname, (1 << 20) * len(comps) / sum(len(c) for c in comps), k, calls, best / n * 1e3, host_ms))
    L.cryo_memrel_destroy(mem)
