/*
 * cryo_codec.h -- C ABI of the MI355X-native cryo-block codec.
 *
 * This is the drop-in boundary underneath pg_cryogen's compression.h
 * (reference compression.h:7-24).  The reference reaches its codec through
 * six third-party calls; each entry point below names the call (file:line in
 * /root/reference) it replaces.  Plain C, plain pointers and sizes, no HIP or
 * torch types.  Nothing here ever calls elog()/exit()/throws: every function
 * returns a cryo_status (0 = ok, negative = error) so that the PG-side C shim
 * (pg_cryogen_amd/host/compression.c) can raise ereport(ERROR) itself without a
 * longjmp crossing C++ frames.
 *
 * Process model: a cryo_codec handle is bound to one GPU and one HIP stream and
 * is used by one thread at a time (a PostgreSQL backend is single-threaded;
 * reference pg_cryogen.c:603-663).  HIP is initialised lazily by
 * cryo_codec_open(), never at library load, so loading the library in the
 * postmaster before fork() is safe.
 *
 * There is NO CPU fallback in this library: without a usable GPU,
 * cryo_codec_open() fails with CRYO_E_NODEV and nothing else can be called.
 */
#ifndef CRYO_CODEC_H
#define CRYO_CODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* numeric values are an on-disk contract: CryoFirstPageHeader.compression_method
 * (reference compression.h:7-11, storage.h:64, cache.c:133) */
enum { CRYO_METHOD_LZ4 = 0, CRYO_METHOD_ZSTD = 1 };

typedef enum {
    CRYO_OK = 0,
    CRYO_E_ARG = -1,         /* bad argument (null pointer, unknown method, size 0 ...) */
    CRYO_E_HIP = -2,         /* HIP runtime call failed (see cryo_codec_last_error)     */
    CRYO_E_NODEV = -3,       /* no usable gfx950 device                                 */
    CRYO_E_CORRUPT = -4,     /* malformed compressed block, or decoded size != block_size */
    CRYO_E_DSTSIZE = -5,     /* destination capacity below cryo_codec_bound()           */
    CRYO_E_UNSUPPORTED = -6, /* valid request this build has no kernel for              */
    CRYO_E_NOMEM = -7
} cryo_status;

typedef struct cryo_codec cryo_codec; /* opaque: device id, stream, workspace */

/* ---- library / device ---- */

/* "cryo-codec X.Y (lz4 block format as liblz4 1.9.3; zstd frames as libzstd 1.4.8)" */
const char *cryo_codec_version(void);
/* number of visible HIP devices, or a negative cryo_status */
int cryo_codec_device_count(void);
/* bind a handle to `device`; creates its stream.  Lazy HIP init happens here. */
int cryo_codec_open(int device, cryo_codec **out);
void cryo_codec_close(cryo_codec *c);
/* text of the last HIP error seen by this handle ("" if none) */
const char *cryo_codec_last_error(const cryo_codec *c);
/* the handle's HIP stream as an opaque pointer (hipStream_t) for profilers/interop */
void *cryo_codec_stream(cryo_codec *c);
/* wait for everything queued on the handle's stream */
int cryo_codec_sync(cryo_codec *c);

/* ---- per-handle options (tuning and tests; the defaults are what production runs) ----
 * Values are read at every call, so a test can flip a path between two batches of one handle. */
typedef enum {
    /* LZ4 decode path: 0 = automatic (by the work in the batch), 1 = in-wave parse kernel (k_lz4_dec_ring),
     * 2 = sequence index + indexed decoder (k_lz4_index* + k_lz4_dec_seq) whatever the batch size, 3 = the few-blocks path
     * (every output byte in parallel, k_lat_*; calls it is not made for -- more than 64 blocks or 64 MiB, blocks below
     * 32 KiB or above 2 MiB -- take the automatic choice) */
    CRYO_OPT_LZ4_DECODE_PATH = 1,
    /* walkers per block of the sequence-index pass: 0 = automatic, else a power of two 1..64 */
    CRYO_OPT_LZ4_INDEX_WALKERS = 2,
    /* K-block host calls: minimum bytes of a call that is cut into pipelined chunks (default 64 MiB) */
    CRYO_OPT_PIPE_MIN_BYTES = 3,
    /* device-resident block pool (cryo_codec_decompress_blocks_keyed): capacity in bytes (0 = pool off, the default) */
    CRYO_OPT_POOL_BYTES = 4,
    /* zstd decode path: 0 = automatic (the four-kernel pipeline; the fused kernel for frames its planner does not take),
     * 1 = the fused one-wave-per-frame kernel for everything, 2 = the pipeline, 3 = the pipeline without the byte-parallel
     * execution it uses for calls of up to 64 frames (k_zlat_* + lat_copy.h): one wave per frame (k_zexec) whatever the call */
    CRYO_OPT_ZSTD_DECODE_PATH = 5,
    /* device workspace kept between calls (the LZ4 sequence index: 2.1 GB for 65 536 x 128 KiB blocks; the zstd decode
     * tiles: up to 12.8 GiB each, four in flight): the host-buffer calls (cryo_codec_*_blocks*), which end synchronised,
     * give back what exceeds this many bytes when they return; -1 = keep everything (the default of a bare handle: a
     * benchmark loop must not reallocate; host/compression.c sets pg_cryogen.gpu_workspace_keep_mb, default 1 GiB) */
    CRYO_OPT_WORKSPACE_KEEP_BYTES = 6,
    /* the most device workspace one call may allocate: 0 = automatic (70 % of what hipMemGetInfo reports free plus what
     * the handle already holds); the zstd decode pipeline runs fewer tiles at once to fit (1 tile at least) */
    CRYO_OPT_WORKSPACE_MAX_BYTES = 7,
    /* 1 (default): the staging worker threads, the pinned staging buffers and -- for the duration of a K-block call of
     * 8 MiB or more -- the calling thread are placed on the cpus of the NUMA node the handle's GPU hangs on (sysfs
     * local_cpulist of its PCI function, cut to the process's affinity mask; the caller's mask is restored on return);
     * 0: wherever the scheduler puts them.  get_option reports 0 when the node could not be determined. */
    CRYO_OPT_NUMA_LOCAL = 8,
    /* waves per block of the indexed LZ4 decoder: 0 = automatic (two for batches that leave most of the chip idle: up to
     * 3 072 blocks; one otherwise), 1 = k_lz4_dec_seq, 2 = k_lz4_dec_dual whatever the batch size */
    CRYO_OPT_LZ4_DECODE_WAVES = 9
} cryo_option;
int cryo_codec_set_option(cryo_codec *c, int option, int64_t value);
/* a long-lived backend between bursts: waits for the handle's queued work, then frees its device workspace, the device and
 * pinned staging buffers of the host-buffer calls and the single-block scratch (they are grow-only otherwise and come
 * back with the next call that needs them).  The device-resident pool stays (CRYO_OPT_POOL_BYTES = 0 frees it).
 * Reference contrast: the CPU libraries hold nothing between calls (compression.c:70-72,102-104 are one-shot). */
int cryo_codec_trim(cryo_codec *c);
int cryo_codec_get_option(const cryo_codec *c, int option, int64_t *value);

/* ---- sizes ---- */

/* replaces LZ4_compressBound (compression.c:67) / ZSTD_compressBound (compression.c:99):
 * identical values (131602 / 131584 at 128 KiB, 1052704 / 1052672 at 1 MiB). 0 on bad args. */
size_t cryo_codec_bound(int method, size_t block_size);

/* ---- device memory plumbing (plain hipMalloc/hipMemcpy wrappers so that C
 *      callers and ctypes need no HIP headers) ----
 * The kernels read compressed input in aligned 16-byte pieces: up to 15 bytes before a block's
 * first byte and after its last byte may be read (never used).  cryo_dev_alloc pads every
 * allocation by 64 bytes; a caller that brings its own device memory must leave that slack
 * after the last compressed block. */
int cryo_dev_alloc(cryo_codec *c, size_t bytes, void **d_ptr);
int cryo_dev_free(cryo_codec *c, void *d_ptr);
int cryo_dev_upload(cryo_codec *c, void *d_dst, const void *h_src, size_t bytes);   /* sync */
int cryo_dev_download(cryo_codec *c, void *h_dst, const void *d_src, size_t bytes); /* sync */
int cryo_dev_memset(cryo_codec *c, void *d_dst, int value, size_t bytes);           /* async */

/* ---- batch codec on DEVICE-RESIDENT buffers (the hot path) ----
 *
 * One wavefront per cryo block; blocks are independent (the reference uses the
 * stateless one-shot APIs, compression.c:70-72,102-104).  All calls are
 * asynchronous on the handle's stream; per-block results land in the device
 * arrays d_out_size / d_status (cryo_status values).
 */

/*
 * Compress n_blocks blocks of block_size bytes.  Block i is read at
 * d_src + i*src_stride and written at d_dst + i*dst_stride
 * (dst_stride >= cryo_codec_bound(method, block_size)).
 *   method LZ4 : replaces LZ4_compress_fast(src,dst,B,LZ4_compressBound(B),accel)
 *                (compression.c:70-72); param = lz4_acceleration_guc (0..50);
 *                output bytes identical to liblz4 1.9.3.
 *   method ZSTD: replaces ZSTD_compress(dst,bound,src,B,level) (compression.c:102-104);
 *                param = zstd_compression_level_guc (-5..22).  Output bytes identical to
 *                libzstd 1.4.8 at every level (strategies fast ... btlazy2 and the optimal parsers
 *                btopt, btultra, btultra2) and every block size; a level above 22 returns
 *                CRYO_E_UNSUPPORTED (no CPU fallback).
 */
int cryo_codec_compress_batch(cryo_codec *c, int method, int param,
                              const void *d_src, uint64_t src_stride,
                              uint32_t block_size, uint64_t n_blocks,
                              void *d_dst, uint64_t dst_stride,
                              uint32_t *d_out_size, int32_t *d_status);

/*
 * Decompress n_blocks blocks.  Compressed block i is the d_src_size[i] bytes at
 * d_src + d_src_off[i]; it must decode to exactly block_size bytes, written at
 * d_dst + i*dst_stride.  d_status[i] = CRYO_OK or CRYO_E_CORRUPT.
 *   method LZ4 : replaces LZ4_decompress_safe(src,dst,csize,B) (compression.c:84)
 *   method ZSTD: replaces ZSTD_decompress(dst,B,src,csize)     (compression.c:116)
 * A stream that decodes to fewer than block_size bytes is reported as
 * CRYO_E_CORRUPT (the reference only Assert()s this, compression.c:88,120).
 */
int cryo_codec_decompress_batch(cryo_codec *c, int method,
                                const void *d_src, const uint64_t *d_src_off,
                                const uint32_t *d_src_size,
                                void *d_dst, uint64_t dst_stride,
                                uint32_t block_size, uint64_t n_blocks,
                                int32_t *d_status);

/* ---- single block, HOST buffers: what cryo_compress()/cryo_decompress()
 *      (compression.c:125-159) call.  Synchronous: H2D, kernel, D2H. ---- */
int cryo_codec_compress_block(cryo_codec *c, int method, int param,
                              const void *h_src, size_t block_size,
                              void *h_dst, size_t dst_cap, size_t *out_size);
int cryo_codec_decompress_block(cryo_codec *c, int method,
                                const void *h_src, size_t src_size,
                                void *h_dst, size_t block_size);

/* ---- K blocks at once, HOST buffers: what the batch write/read staging calls
 *      (write-behind of K full blocks from multi_insert, read-ahead of K block chains;
 *      reference one-at-a-time equivalents: pg_cryogen.c:726 and cache.c:178).
 *      Synchronous: one H2D, one kernel launch, one D2H for the whole batch; device buffers and a pinned
 *      staging buffer are kept in the handle (grow-only). ---- */
/* block i: h_src + i*block_size  ->  h_dst + i*dst_stride, size in h_out_size[i].  dst_stride >= bound;
 * the whole slot (up to bound bytes) may be written, only the first h_out_size[i] bytes are meaningful */
int cryo_codec_compress_blocks(cryo_codec *c, int method, int param,
                               const void *h_src, size_t block_size, size_t n_blocks,
                               void *h_dst, size_t dst_stride, uint32_t *h_out_size);
/* block i: h_src[i] (h_src_size[i] bytes) -> h_dst + i*block_size; h_status[i] = CRYO_OK / CRYO_E_CORRUPT.
 * Returns CRYO_OK when the batch ran, even if some blocks are corrupt. */
int cryo_codec_decompress_blocks(cryo_codec *c, int method,
                                 const void *const *h_src, const uint32_t *h_src_size, size_t n_blocks,
                                 void *h_dst, size_t block_size, int32_t *h_status);
/* the same with one destination pointer per block (the slots of the decompressed-block cache, reference
 * cache.c:46,178: `out` is a cache entry's data[]): block i -> h_dst[i]; a block whose status is not CRYO_OK
 * leaves its destination untouched */
int cryo_codec_decompress_blocks_to(cryo_codec *c, int method,
                                    const void *const *h_src, const uint32_t *h_src_size, size_t n_blocks,
                                    void *const *h_dst, size_t block_size, int32_t *h_status);

/* ---- device-resident block pool ("optional device-resident compressed/decompressed pool so
 *      repeated scans skip PCIe"; the reference's cache is host-only: cache.c:17-50).
 *      With CRYO_OPT_POOL_BYTES > 0 the decoded blocks of keyed calls stay in HBM (first in, first out).  A key is the
 *      caller's identity of a block -- the host cache passes (relation oid << 32 | first block number), the key of
 *      reference cache.c:37-47 -- and 0 means "do not keep".  A block found in the pool with the same compressed size and
 *      the same 64-bit hash of its whole compressed stream is copied back from HBM: nothing travels towards the
 *      device and no kernel decodes it.  A relation that is rewritten or truncated must be dropped with
 *      cryo_codec_pool_invalidate (reference: the relcache callback, pg_cryogen.c:163-167). ---- */
int cryo_codec_decompress_blocks_keyed(cryo_codec *c, int method, const uint64_t *keys,
                                       const void *const *h_src, const uint32_t *h_src_size, size_t n_blocks,
                                       void *const *h_dst, size_t block_size, int32_t *h_status);
/* drop the entries whose key's upper 32 bits equal key_hi; all entries when all_entries != 0 */
int cryo_codec_pool_invalidate(cryo_codec *c, uint32_t key_hi, int all_entries);
typedef struct {
    uint64_t h2d_bytes, d2h_bytes;       /* bytes the host-buffer calls moved across PCIe, each direction */
    uint64_t pool_hits, pool_misses;     /* keyed blocks served from HBM / decoded                       */
    uint64_t pool_blocks, pool_capacity; /* blocks held now / slots                                      */
} cryo_codec_transfer_counters;
int cryo_codec_get_transfer_counters(const cryo_codec *c, cryo_codec_transfer_counters *out);

/* ---- several GPUs behind one call: the dispatcher of BASELINE's "independent cryo blocks from a COPY multi_insert
 *      or a seq-scan shard embarrassingly across the 8 GPUs of one node (round-robin dispatch, no collective)".
 *      One codec handle per listed device (a device may be listed more than once), block i of a call goes to
 *      handle i mod G, one host thread per handle drives its share through the K-block calls above; results land
 *      where the single-handle calls would have put them.  The reference has no counterpart (one block, one core:
 *      pg_cryogen.c:726, cache.c:178). ---- */
typedef struct cryo_multi cryo_multi;
int cryo_multi_open(const int *devices, int n_devices, cryo_multi **out);
void cryo_multi_close(cryo_multi *m);
int cryo_multi_count(const cryo_multi *m);
const char *cryo_multi_last_error(const cryo_multi *m);
int cryo_multi_compress_blocks(cryo_multi *m, int method, int param,
                               const void *h_src, size_t block_size, size_t n_blocks,
                               void *h_dst, size_t dst_stride, uint32_t *h_out_size);
int cryo_multi_decompress_blocks(cryo_multi *m, int method,
                                 const void *const *h_src, const uint32_t *h_src_size, size_t n_blocks,
                                 void *h_dst, size_t block_size, int32_t *h_status);
/* one destination per block (cryo_codec_decompress_blocks_to across the devices): what the decompressed-block cache
 * binds, its slots are the destinations (reference cache.c:46,178) */
int cryo_multi_decompress_blocks_to(cryo_multi *m, int method,
                                    const void *const *h_src, const uint32_t *h_src_size, size_t n_blocks,
                                    void *const *h_dst, size_t block_size, int32_t *h_status);
/* keyed (pool) variant: a keyed block goes to handle (key mod G), so that it finds its pool entry again; options and
 * invalidation reach every handle; the counters are summed */
int cryo_multi_decompress_blocks_keyed(cryo_multi *m, int method, const uint64_t *keys,
                                       const void *const *h_src, const uint32_t *h_src_size, size_t n_blocks,
                                       void *const *h_dst, size_t block_size, int32_t *h_status);
int cryo_multi_set_option(cryo_multi *m, int option, int64_t value);
int cryo_multi_pool_invalidate(cryo_multi *m, uint32_t key_hi, int all_entries);
int cryo_multi_trim(cryo_multi *m);
int cryo_multi_get_transfer_counters(const cryo_multi *m, cryo_codec_transfer_counters *out);

/* ---- batch helpers used by staging, tests and the benchmark ---- */

/* synthetic cryo blocks (include/cryo_synth.h) on device: slot k holds job block
 * first_block + k*block_step (block_step = N, first_block = rank gives rank's round-robin share) */
int cryo_codec_synth_batch(cryo_codec *c, uint64_t seed, uint64_t first_block, uint64_t block_step,
                           uint64_t n_blocks, uint32_t block_size, int dist, void *d_dst, uint64_t dst_stride);
/* per-block 64-bit checksum (same function as cryo_checksum64() below) */
int cryo_codec_checksum_batch(cryo_codec *c, const void *d_src, uint64_t src_stride,
                              const uint32_t *d_sizes /* or NULL: fixed_size */, uint32_t fixed_size,
                              uint64_t n_blocks, uint64_t *d_sums);
/* number of blocks whose bytes differ between two batches -> *d_mismatch (device u64, accumulated) */
int cryo_codec_compare_batch(cryo_codec *c, const void *d_a, uint64_t a_stride,
                             const void *d_b, uint64_t b_stride, uint32_t block_size,
                             uint64_t n_blocks, uint64_t *d_mismatch);
/* host-side reference of the checksum, for tests */
uint64_t cryo_checksum64(const void *p, size_t n);

/* ---- timing on the handle's stream (HIP events) ---- */
int cryo_codec_timer_start(cryo_codec *c);
/* records the end event, waits for it, returns elapsed milliseconds */
int cryo_codec_timer_stop(cryo_codec *c, float *ms);

/* ---- counters (SURVEY.md section 5 "metrics") ---- */
typedef struct {
    uint64_t blocks_compressed, blocks_decompressed;
    uint64_t bytes_in, bytes_out;   /* uncompressed bytes through compress / decompress */
    uint64_t launches;
} cryo_codec_counters;
int cryo_codec_get_counters(const cryo_codec *c, cryo_codec_counters *out);

#ifdef __cplusplus
}
#endif
#endif /* CRYO_CODEC_H */
