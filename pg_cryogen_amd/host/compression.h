/*
 * compression.h -- host-side mirror of the reference's codec boundary
 * (reference compression.h:7-24): same type, same three GUC variables, same three
 * functions with the same argument meaning, ownership and error behaviour
 * (SURVEY.md section 8b).  pg_cryogen.c:726 and cache.c:178 call it unchanged.
 *
 * Underneath, instead of liblz4/libzstd, it drives the MI355X codec through the C ABI
 * (include/cryo_codec.h).
 */
#ifndef __COMPRESSION_H__
#define __COMPRESSION_H__

#ifdef CRYO_HAVE_POSTGRES
#include "postgres.h"
#else
#include "pg_compat.h"
#endif

typedef enum
{
    COMP_LZ4 = 0,
    COMP_ZSTD
} CompressionMethod;

extern int compression_method_guc;
extern int lz4_acceleration_guc;
extern int zstd_compression_level_guc;

extern char *cryo_compress(CompressionMethod method,
                           const char *data,
                           Size *compressed_size);
extern bool cryo_decompress(CompressionMethod method,
                            const char *compressed,
                            Size compressed_size,
                            char *out);
extern void cryo_define_compression_gucs(void);

/* ---- additions (not in the reference) ---- */

/* the reference hard-codes CRYO_BLCKSZ = 1 MiB (storage.h:18); here it is a run-time value
 * with the same default so the benchmark's 128 KiB blocks use the same code */
extern Size cryo_blcksz;
/* GPU used by this backend (additive GUC pg_cryogen.gpu_device, default 0) */
extern int cryo_gpu_device_guc;
/* how many GPUs, from gpu_device on, the K-block calls are spread over (additive GUC pg_cryogen.gpu_count, default 1) */
extern int cryo_gpu_count_guc;
/* device-resident pool of decoded blocks, MiB over all GPUs of the backend (additive GUC pg_cryogen.gpu_pool_mb, default 0 = off) */
extern int cryo_gpu_pool_mb_guc;
extern int cryo_gpu_workspace_keep_mb_guc; /* pg_cryogen.gpu_workspace_keep_mb (default 1024, -1 = keep everything) */
extern int cryo_gpu_readahead_blocks_guc;  /* pg_cryogen.gpu_readahead_blocks (default 8, 1 = off): host/cache.c, cryo_read_data_rel */
/* bytes the codec moved towards the device / back, blocks served from the pool / decoded (0 when no GPU codec is bound) */
void cryo_host_transfer_counters(uint64_t *h2d_bytes, uint64_t *d2h_bytes, uint64_t *pool_hits, uint64_t *pool_misses);

/* the codec entry points the host side calls; production binds them to libcryo_codec.so
 * (include/cryo_codec.h), CPU-only plumbing tests may bind a test double */
typedef struct CryoCodecOps {
    size_t (*bound)(int method, size_t block_size);
    int (*compress_blocks)(void *ctx, int method, int param, const void *src, size_t block_size, size_t n,
                           void *dst, size_t dst_stride, uint32_t *out_size);
    int (*decompress_blocks)(void *ctx, int method, const void *const *src, const uint32_t *src_size, size_t n,
                             void *dst, size_t block_size, int32_t *status);
    void *ctx;
    /* optional (may be NULL): one destination per block, so the cache decodes straight into its slots */
    int (*decompress_blocks_scatter)(void *ctx, int method, const void *const *src, const uint32_t *src_size, size_t n,
                                     void *const *dst, size_t block_size, int32_t *status);
    /* optional (may be NULL): the same with a key per block (relation oid << 32 | first block number): a codec with a
     * device-resident pool (pg_cryogen.gpu_pool_mb) serves a block it still holds from HBM -- nothing crosses PCIe towards
     * the device, no kernel runs for it */
    int (*decompress_blocks_keyed)(void *ctx, int method, const uint64_t *keys, const void *const *src, const uint32_t *src_size,
                                   size_t n, void *const *dst, size_t block_size, int32_t *status);
    /* optional (may be NULL): forget the pooled blocks of a relation (reference: relcache callback, pg_cryogen.c:163-167) */
    void (*pool_invalidate)(void *ctx, uint32_t relid);
} CryoCodecOps;
#ifdef CRYO_HOST_TEST_HOOKS
void cryo_host_set_codec_ops(const CryoCodecOps *ops); /* test builds only: bind a double; NULL restores the HIP binding */
#endif
const CryoCodecOps *cryo_host_codec_ops(void);         /* lazily opens the GPU codec */
void cryo_host_codec_trim(void);                         /* idle backend: free the binding's device workspace and staging buffers */
size_t cryo_host_codec_bound(int method, size_t n);      /* cryo_codec_bound (or the bound double's): never opens the GPU */
const CryoCodecOps *cryo_host_codec_ops_if_open(void); /* the binding if there is one already; never opens the GPU */
const char *cryo_host_codec_error(void);

#endif /* __COMPRESSION_H__ */
