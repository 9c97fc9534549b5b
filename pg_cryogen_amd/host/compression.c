/*
 * compression.c -- cryo_compress / cryo_decompress over the MI355X codec C ABI.
 *
 * Mirrors reference compression.c:16-159: the three GUCs with the same names, ranges and
 * defaults (compression.c:24-58), palloc(bound) output owned by the caller
 * (pg_cryogen.c:826 pfrees it), elog(ERROR, "pg_cryogen: compression failed") on failure
 * (compression.c:73-74,105-106), `false` from cryo_decompress on malformed input
 * (compression.c:85-86,117-118), elog(ERROR) on an unknown method (compression.c:137,157).
 * Differences, all invisible to valid data: a block that decodes to fewer than CRYO_BLCKSZ
 * bytes returns false (the reference only Assert()s, compression.c:88,120).
 *
 * The GPU codec is opened lazily on first use, never in _PG_init: a library preloaded by the
 * postmaster must not create a HIP context before fork().  The C ABI never
 * longjmps; elog(ERROR) is raised here, in C, after the call returned.
 */
#include "compression.h"
#include "cryo_codec.h"
#ifdef CRYO_HAVE_POSTGRES
#include "utils/guc.h" /* DefineCustom*Variable, config_enum_entry, PGC_USERSET (reference compression.c:4) */
#endif

#include <stdarg.h>
#include <stdio.h>
#include <time.h>

int compression_method_guc = COMP_ZSTD;
int lz4_acceleration_guc = 1;
int zstd_compression_level_guc = 1;
int cryo_gpu_device_guc = 0;
int cryo_gpu_count_guc = 1;
int cryo_gpu_pool_mb_guc = 0;
int cryo_gpu_workspace_keep_mb_guc = 1024; /* device workspace a backend keeps between calls (-1: everything) */
int cryo_gpu_readahead_blocks_guc = 8;    /* cryo blocks a sequential scan's cache miss decodes with one codec call (1: only the block asked for) */
Size cryo_blcksz = (Size)1 << 20; /* CRYO_BLCKSZ, reference storage.h:18 */

/* ---------------- codec binding ---------------- */
/* one handle per GPU of this backend: pg_cryogen.gpu_device is the first, pg_cryogen.gpu_count how many (consecutive
 * device numbers, wrapping).  With more than one, the K-block calls of the staging and cache code go through the
 * dispatcher of include/cryo_codec.h (block i of a call -> GPU i mod G, one host thread per GPU). */
static cryo_multi *hip_multi;
static int hip_multi_first = -1, hip_multi_count = 0, hip_pool_mb = 0, hip_keep_mb = -2;
/* A failed open is remembered until the GUCs change -- for good when the machine has no GPU (deterministic), for
 * CRYO_OPEN_RETRY_SECONDS when devices exist but cryo_multi_open failed (out of device memory, a busy device: transient;
 * a pooled backend must not refuse every cryo table for the rest of its life because of one bad moment). */
#define CRYO_OPEN_RETRY_SECONDS 5
static int hip_open_failed, hip_failed_first = -1, hip_failed_count = 0;
static time_t hip_failed_at; /* 0: the failure was deterministic */
static char codec_err[320];

static size_t hip_bound(int method, size_t n) { return cryo_codec_bound(method, n); }
static int hip_compress_blocks(void *ctx, int method, int param, const void *src, size_t bs, size_t n, void *dst,
                               size_t stride, uint32_t *out)
{
    return cryo_multi_compress_blocks((cryo_multi *)ctx, method, param, src, bs, n, dst, stride, out);
}
static int hip_decompress_blocks(void *ctx, int method, const void *const *src, const uint32_t *sz, size_t n,
                                 void *dst, size_t bs, int32_t *st)
{
    return cryo_multi_decompress_blocks((cryo_multi *)ctx, method, src, sz, n, dst, bs, st);
}

/* one destination per block: the decompressed-block cache hands its slots over (reference cache.c:46,178) */
static int hip_decompress_blocks_scatter(void *ctx, int method, const void *const *src, const uint32_t *sz, size_t n,
                                         void *const *dst, size_t bs, int32_t *st)
{
    return cryo_multi_decompress_blocks_to((cryo_multi *)ctx, method, src, sz, n, dst, bs, st);
}

static int hip_decompress_blocks_keyed(void *ctx, int method, const uint64_t *keys, const void *const *src, const uint32_t *sz,
                                       size_t n, void *const *dst, size_t bs, int32_t *st)
{
    return cryo_multi_decompress_blocks_keyed((cryo_multi *)ctx, method, keys, src, sz, n, dst, bs, st);
}
/* relid 0 (InvalidOid): PostgreSQL's relcache callback after a sinval-queue reset -- "anything may have changed": every entry */
static void hip_pool_invalidate(void *ctx, uint32_t relid) { (void)cryo_multi_pool_invalidate((cryo_multi *)ctx, relid, relid == 0u); }

static CryoCodecOps hip_ops = {hip_bound, hip_compress_blocks, hip_decompress_blocks, NULL, hip_decompress_blocks_scatter,
                               hip_decompress_blocks_keyed, hip_pool_invalidate};
static const CryoCodecOps *bound_ops; /* CPU-only plumbing tests bind a double here (CRYO_HOST_TEST_HOOKS builds only) */

#ifdef CRYO_HOST_TEST_HOOKS
void cryo_host_set_codec_ops(const CryoCodecOps *ops) { bound_ops = ops; }
#endif
const char *cryo_host_codec_error(void) { return codec_err; }

const CryoCodecOps *cryo_host_codec_ops(void)
{
    if (bound_ops) return bound_ops;
    if (hip_multi && (hip_multi_first != cryo_gpu_device_guc || hip_multi_count != cryo_gpu_count_guc)) {
        cryo_multi_close(hip_multi); /* the GUCs changed: rebind */
        hip_multi = NULL;
    }
    if (!hip_multi) {
        int devs[64], i, ndev, rc;
        int cnt = cryo_gpu_count_guc < 1 ? 1 : (cryo_gpu_count_guc > 64 ? 64 : cryo_gpu_count_guc);
        /* not again for every call of a backend that cannot have a GPU (codec_err still says why) */
        if (hip_open_failed && hip_failed_first == cryo_gpu_device_guc && hip_failed_count == cryo_gpu_count_guc &&
            (hip_failed_at == 0 || time(NULL) - hip_failed_at < CRYO_OPEN_RETRY_SECONDS))
            return NULL;
        hip_open_failed = 1;
        hip_failed_at = 0;
        hip_failed_first = cryo_gpu_device_guc;
        hip_failed_count = cryo_gpu_count_guc;
        ndev = cryo_codec_device_count();
        if (ndev <= 0) {
            snprintf(codec_err, sizeof codec_err, "no GPU visible to the HIP runtime (%d; no CPU fallback)", ndev);
            return NULL;
        }
        for (i = 0; i < cnt; i++) devs[i] = (cryo_gpu_device_guc + i) % ndev;
        rc = cryo_multi_open(devs, cnt, &hip_multi);
        if (rc != CRYO_OK) {
            snprintf(codec_err, sizeof codec_err, "cryo_multi_open(first %d, count %d) failed with %d (no CPU fallback)",
                     cryo_gpu_device_guc, cnt, rc);
            hip_multi = NULL;
            hip_failed_at = time(NULL); /* devices exist: try again in a while */
            if (hip_failed_at == 0) hip_failed_at = 1;
            return NULL;
        }
        hip_open_failed = 0;
        hip_multi_first = cryo_gpu_device_guc;
        hip_multi_count = cryo_gpu_count_guc;
        hip_pool_mb = 0;
        hip_keep_mb = -2;
        hip_ops.ctx = hip_multi;
    }
    if (hip_keep_mb != cryo_gpu_workspace_keep_mb_guc) { /* a backend is long-lived: one large call must not pin its workspace for good */
        hip_keep_mb = cryo_gpu_workspace_keep_mb_guc;
        (void)cryo_multi_set_option(hip_multi, CRYO_OPT_WORKSPACE_KEEP_BYTES, hip_keep_mb < 0 ? -1 : (int64_t)hip_keep_mb << 20);
    }
    if (hip_pool_mb != cryo_gpu_pool_mb_guc) { /* the GUC changed: resize (0 frees the pool) */
        hip_pool_mb = cryo_gpu_pool_mb_guc;
        (void)cryo_multi_set_option(hip_multi, CRYO_OPT_POOL_BYTES, (int64_t)(hip_pool_mb < 0 ? 0 : hip_pool_mb) << 20);
    }
    return &hip_ops;
}

/* worst-case compressed size of a block: pure arithmetic (cryo_codec_bound), no GPU -- the cache sizes its chain lists with
 * it at _PG_init (cryo_init_cache, reference pg_cryogen.c:172), long before a backend may ever touch a cryo table */
size_t cryo_host_codec_bound(int method, size_t n)
{
    if (bound_ops) return bound_ops->bound(method, n);
    return cryo_codec_bound(method, n);
}

/* The binding if this backend has one already -- never opens the GPU.  For the relcache invalidation callback
 * (cryo_cache_invalidate_relation), which PostgreSQL fires for every invalidation of any relation in every backend that
 * loaded the extension: a backend that never touched a cryo table must not pay a HIP context for a sinval message. */
const CryoCodecOps *cryo_host_codec_ops_if_open(void)
{
    if (bound_ops) return bound_ops;
    return hip_multi ? &hip_ops : NULL;
}

void cryo_host_transfer_counters(uint64_t *h2d_bytes, uint64_t *d2h_bytes, uint64_t *pool_hits, uint64_t *pool_misses)
{
    cryo_codec_transfer_counters t = {0, 0, 0, 0, 0, 0};
    if (hip_multi) (void)cryo_multi_get_transfer_counters(hip_multi, &t);
    if (h2d_bytes) *h2d_bytes = t.h2d_bytes;
    if (d2h_bytes) *d2h_bytes = t.d2h_bytes;
    if (pool_hits) *pool_hits = t.pool_hits;
    if (pool_misses) *pool_misses = t.pool_misses;
}

/* idle backend: give the GPU memory of the binding back (device workspace, staging buffers; not the pool) */
void cryo_host_codec_trim(void)
{
    if (hip_multi) (void)cryo_multi_trim(hip_multi);
}

/* ---------------- GUCs ---------------- */
void cryo_define_compression_gucs(void)
{
#ifdef CRYO_HAVE_POSTGRES
    static const struct config_enum_entry compression_method_options[] = {
        {"lz4", COMP_LZ4, false}, {"zstd", COMP_ZSTD, false}, {NULL, 0, false}};
    DefineCustomEnumVariable("pg_cryogen.compression_method", "Possible values are lz4 and zstd.", NULL,
                             &compression_method_guc, COMP_ZSTD, compression_method_options, PGC_USERSET, 0,
                             NULL, NULL, NULL);
    DefineCustomIntVariable("pg_cryogen.lz4_acceleration", "Sets lz4 acceleration.", NULL,
                            &lz4_acceleration_guc, 1, 0, 50, PGC_USERSET, 0, NULL, NULL, NULL);
    DefineCustomIntVariable("pg_cryogen.zstd_compression_level", "Sets zstd compression level.", NULL,
                            &zstd_compression_level_guc, 1, -5, 22, PGC_USERSET, 0, NULL, NULL, NULL);
    DefineCustomIntVariable("pg_cryogen.gpu_device", "GPU used by this backend.", NULL, &cryo_gpu_device_guc, 0,
                            0, 63, PGC_USERSET, 0, NULL, NULL, NULL);
    DefineCustomIntVariable("pg_cryogen.gpu_count", "Number of GPUs (from gpu_device on) the K-block calls of this backend are spread over.",
                            NULL, &cryo_gpu_count_guc, 1, 1, 64, PGC_USERSET, 0, NULL, NULL, NULL);
    DefineCustomIntVariable("pg_cryogen.gpu_pool_mb", "Decoded blocks kept in GPU memory so that repeated scans skip the transfer and the decode (MiB, 0 = off).",
                            NULL, &cryo_gpu_pool_mb_guc, 0, 0, 262144, PGC_USERSET, 0, NULL, NULL, NULL);
    DefineCustomIntVariable("pg_cryogen.gpu_workspace_keep_mb", "GPU workspace this backend keeps between codec calls (MiB; -1 = everything a call ever needed).",
                            NULL, &cryo_gpu_workspace_keep_mb_guc, 1024, -1, 262144, PGC_USERSET, 0, NULL, NULL, NULL);
    DefineCustomIntVariable("pg_cryogen.gpu_readahead_blocks",
                            "Cryo blocks a sequential scan's cache miss decodes with one codec call (1 = only the block asked for).",
                            NULL, &cryo_gpu_readahead_blocks_guc, 8, 1, 64, PGC_USERSET, 0, NULL, NULL, NULL);
#else
    /* no GUC machinery without PostgreSQL: the variables keep the reference's defaults */
    compression_method_guc = COMP_ZSTD;
    lz4_acceleration_guc = 1;
    zstd_compression_level_guc = 1;
#endif
}

/* ---------------- the two calls ---------------- */
static int method_param(CompressionMethod method)
{
    return method == COMP_LZ4 ? lz4_acceleration_guc : zstd_compression_level_guc;
}

char *cryo_compress(CompressionMethod method, const char *data, Size *compressed_size)
{
    const CryoCodecOps *ops;
    Size estimate;
    char *compressed;
    uint32_t csize = 0;
    int rc;

    if (method != COMP_LZ4 && method != COMP_ZSTD)
        elog(ERROR, "pg_cryogen: unknown compression method");
    ops = cryo_host_codec_ops();
    if (!ops)
        elog(ERROR, "pg_cryogen: compression failed (%s)", codec_err);
    estimate = ops->bound((int)method, cryo_blcksz);
    compressed = palloc(estimate);
    rc = ops->compress_blocks(ops->ctx, (int)method, method_param(method), data, cryo_blcksz, 1, compressed,
                              estimate, &csize);
    if (rc != 0 || csize == 0) {
        pfree(compressed);
        if (rc == CRYO_E_UNSUPPORTED)   /* same ERROR as the reference, with the reason (this build has no kernel for it) */
            elog(ERROR, "pg_cryogen: compression failed (no GPU kernel for %s parameter %d at block size %lu)",
                 method == COMP_LZ4 ? "lz4" : "zstd", method_param(method), (unsigned long)cryo_blcksz);
        else
            elog(ERROR, "pg_cryogen: compression failed");
        return NULL;
    }
    *compressed_size = csize;
    return compressed;
}

bool cryo_decompress(CompressionMethod method, const char *compressed, Size compressed_size, char *out)
{
    const CryoCodecOps *ops;
    const void *srcs[1];
    uint32_t sizes[1];
    int32_t status[1] = {0};
    int rc;

    if (method != COMP_LZ4 && method != COMP_ZSTD)
        elog(ERROR, "pg_cryogen: unknown compression method");
    if (compressed_size == 0 || compressed_size > 0xFFFFFFFFu)
        return false;
    ops = cryo_host_codec_ops();
    if (!ops)
        elog(ERROR, "pg_cryogen: decompression unavailable (%s)", codec_err);
    srcs[0] = compressed;
    sizes[0] = (uint32_t)compressed_size;
    rc = ops->decompress_blocks(ops->ctx, (int)method, srcs, sizes, 1, out, cryo_blcksz, status);
    if (rc != 0)
        elog(ERROR, "pg_cryogen: decompression failed to run (%d)", rc);
    return status[0] == 0;
}

#ifndef CRYO_HAVE_POSTGRES
/* ---------------- pg_compat: elog ---------------- */
static cryo_error_handler err_handler;
void cryo_compat_set_error_handler(cryo_error_handler h) { err_handler = h; }
void cryo_compat_elog(int elevel, const char *fmt, ...)
{
    char msg[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof msg, fmt, ap);
    va_end(ap);
    if (err_handler) { err_handler(elevel, msg); return; }
    if (elevel >= ERROR) { fprintf(stderr, "ERROR: %s\n", msg); abort(); }
}
#endif
