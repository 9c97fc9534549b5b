/*
 * kernels.h -- internal launcher interface between the C-ABI host code
 * (cryo_codec.cpp) and the gfx950 kernels.  Not installed; the public surface
 * is include/cryo_codec.h.
 */
#ifndef CRYO_KERNELS_H
#define CRYO_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

/* Tuning and debugging switches read from the environment exist in -DCRYO_DEBUG builds only (the A/B builds under
 * profiles/): the product library reads no environment variable but CRYO_HOST_THREADS (cryo_codec.cpp); what a
 * deployment may change is a per-handle option (cryo_codec_set_option). */
static inline const char *cryo_tuning_env(const char *name)
{
#ifdef CRYO_DEBUG
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

namespace cryo {

/* synthetic cryo blocks, include/cryo_synth.h */
hipError_t launch_synth(hipStream_t s, uint64_t seed, uint64_t first_block, uint64_t block_step, uint64_t n_blocks,
                        uint32_t block_size, int dist, uint8_t *d_dst, uint64_t dst_stride);

hipError_t launch_checksum(hipStream_t s, const uint8_t *d_src, uint64_t src_stride,
                           const uint32_t *d_sizes, uint32_t fixed_size, uint64_t n_blocks,
                           uint64_t *d_sums);

hipError_t launch_compare(hipStream_t s, const uint8_t *d_a, uint64_t a_stride, const uint8_t *d_b,
                          uint64_t b_stride, uint32_t block_size, uint64_t n_blocks,
                          uint64_t *d_mismatch);

/* device-resident pool: up to 64 blocks per launch from pool slots (by value: no transfer towards the device) to
 * slots first .. first+count-1 of a contiguous staging area */
struct GatherSlots { uint32_t slot[64]; };
hipError_t launch_gather_blocks(hipStream_t s, const uint8_t *d_pool, const GatherSlots &slots, uint32_t count, uint8_t *d_dst,
                                uint32_t block_size, uint32_t first);

/* LZ4 block format */
constexpr uint32_t kLz4IndexResidentLanes = 65536u;
/* per-handle options (include/cryo_codec.h: CRYO_OPT_LZ4_DECODE_PATH, CRYO_OPT_LZ4_INDEX_WALKERS); 0 = automatic */
struct Lz4DecodeOpts {
    int path = 0;    /* 1: in-wave parse kernel, 2: sequence index + indexed decoder, 3: few blocks, every output byte in parallel (lz4_lat.hip) */
    int walkers = 0; /* walkers per block of the index pass (power of two, 1..64) */
    int waves = 0;   /* waves per block of the indexed decoder: 1 = k_lz4_dec_seq, 2 = k_lz4_dec_dual, 0 = by the batch size */
    /* a low-priority side stream with its events (created by the handle, may be null): the last, partial round of a batch
     * that fills the chip once or a few times is decoded there with two waves per block (lz4_dec2.hip, launch_dec_seq) */
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int cus = 0;     /* compute units of the handle's device (hipDeviceProp.multiProcessorCount; 0: an MI355X's 256) */
};
hipError_t launch_lz4_decompress(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                                 const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                                 uint32_t block_size, uint64_t n_blocks, int32_t *d_status, void *d_workspace,
                                 size_t workspace_bytes, const Lz4DecodeOpts &opts);
/* bytes of workspace the sequence-index pass of this batch wants (0: the batch is decoded without one) */
size_t lz4_decompress_workspace(uint64_t n_blocks, uint32_t block_size, const Lz4DecodeOpts &opts);
/* which path a batch takes: 0 = in-wave parse kernel (k_lz4_dec_ring), else the walkers per block of the index pass */
uint32_t lz4_decode_plan(uint64_t n_blocks, uint32_t block_size, const Lz4DecodeOpts &opts);
/* lz4_index.hip: the sequence index.  Workspace layout: n_blocks rows of `cap` 16-bit entries; a row = S sub-rows of
 * [ext entries: extension of the left neighbour][cap_main entries: the segment's own records]; 64 x 32 bytes of dummy
 * slots; one uint2 descriptor per (block, segment): x = entries used in the extension | first valid own record << 16,
 * y = valid own records */
struct Lz4IndexLayout {
    uint32_t logS, cap_main, ext, cap;
    size_t dummy_off, seg_off, bytes;
};
Lz4IndexLayout lz4_index_layout(uint64_t n_blocks, uint32_t block_size, uint32_t walkers);
hipError_t launch_lz4_index(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off, const uint32_t *d_src_size,
                            uint64_t n_blocks, uint32_t block_size, void *d_workspace, const Lz4IndexLayout &L);
/* few blocks per call: a workgroup of up to 1 024 walkers per block, direct loads (lz4_index.hip, k_lz4_index_few) */
Lz4IndexLayout lz4_index_layout_few(uint64_t n_blocks, uint32_t block_size);
const uint32_t *lz4_index_few_failed(const void *d_workspace, const Lz4IndexLayout &L, uint64_t n_blocks); /* one word per block: 1 = no index */
hipError_t launch_lz4_index_few(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off, const uint32_t *d_src_size,
                                uint64_t n_blocks, uint32_t block_size, void *d_workspace, const Lz4IndexLayout &L);
/* Per-block routing of batches indexed with several walkers per block: a block that compressed to more than 15/16 of
 * its size is almost all literal runs -- a walker that starts at a guessed position inside one hops through it ~7 bytes at
 * a time where the true chain takes one hop, and the copy is one long run either way -- so such blocks are left to the
 * in-wave parser (k_lz4_dec_ring, launched over the same batch with the opposite filter).  Measured at 16 384 x 128 KiB:
 * incompressible blocks 961 GB/s through the index, 2 203 through the parser; acceleration-50 streams (ratio 1.05) 309
 * against 378 at 8 192 blocks; the headline data (ratio 1.29, a token every 16 bytes) 920 against 529
 * (profiles/r03_lz4_decode_batch_shapes.txt). */
__host__ __device__ inline bool lz4_literal_heavy(uint32_t csize, uint32_t block_size) { return csize > block_size - (block_size >> 4); }
/* A stream of less than 16 KiB gets ONE index walker whatever the batch: it is walked in ~0.3 ms at most, and such streams are
 * the highly periodic ones (blocks of fixed-width rows: a token every 3-5 bytes, every sequence alike) on which chains started
 * at guessed positions run beside the true one for ever -- the hand-over fails and the block is walked again by one walker
 * anyway (8 192 x 1 MiB `int4`: 1.0-1.1 ms of index pass; profiles/r05_index_run255.txt).  Used by k_lz4_index, k_lz4_few_* and
 * lz4_lat.hip alike: they must cut a block into the same segments. */
__host__ __device__ inline bool lz4_index_one_walker(uint32_t csize) { return csize < 16384u; }
hipError_t launch_lz4_dec_ring(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off, const uint32_t *d_src_size,
                               uint8_t *d_dst, uint64_t dst_stride, uint32_t block_size, uint64_t n_blocks, int32_t *d_status,
                               bool only_literal_heavy, const uint32_t *d_done = nullptr /* blocks marked there are skipped */);
/* lz4_dec2.hip: index pass + the decoder that consumes it */
hipError_t launch_lz4_decompress_indexed(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                                         const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                                         uint32_t block_size, uint64_t n_blocks, int32_t *d_status, void *d_workspace,
                                         size_t workspace_bytes, uint32_t walkers, int waves = 0, const Lz4DecodeOpts *opts = nullptr);

/* lz4_lat.hip: few blocks per call (the reference's own call shapes) */
bool lz4_latency_eligible(uint64_t n_blocks, uint32_t block_size);
size_t lz4_latency_workspace(uint64_t n_blocks, uint32_t block_size);
hipError_t launch_lz4_decompress_latency(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                                         const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                                         uint32_t block_size, uint64_t n_blocks, int32_t *d_status, void *d_workspace,
                                         size_t workspace_bytes);

hipError_t launch_lz4_compress(hipStream_t s, const uint8_t *d_src, uint64_t src_stride,
                               uint32_t block_size, uint64_t n_blocks, uint8_t *d_dst,
                               uint64_t dst_stride, int accel, uint32_t *d_out_size,
                               int32_t *d_status);

hipError_t launch_lz4_compress_batch64(hipStream_t s, const uint8_t *d_src, uint64_t src_stride,
                                       uint32_t block_size, uint64_t n_blocks, uint8_t *d_dst,
                                       uint64_t dst_stride, int accel, uint32_t *d_out_size, int32_t *d_status);

/* zstd frames.  `aux` (optional): two side streams + events the batch pipeline alternates its tiles on;
 * the work is ordered after everything already queued on `s`, and `s` waits for it before returning. */
constexpr int kZstdLanes = 8; /* side streams a decode call may spread its tiles over */
struct ZstdAux {
    hipStream_t lane[kZstdLanes];
    hipEvent_t fork, join[kZstdLanes];
    /* inside a tile the Huffman stage (k_zhufw, k_zmove, k_zhuf) and the sequence stage (k_zchain4, k_zmat) depend on
     * k_zplan only and meet in k_zexec: calls of few tiles run them side by side (round 5) */
    hipStream_t side[kZstdLanes];
    hipEvent_t planned[kZstdLanes], seqs_done[kZstdLanes];
};
hipError_t launch_zstd_decompress(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                                  const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                                  uint32_t block_size, uint64_t n_blocks, int32_t *d_status,
                                  void *d_workspace, size_t workspace_bytes, const ZstdAux *aux, int path);
/* max_bytes: the most the call may use; the pipeline plans fewer tiles in flight to fit (one at least) */
size_t zstd_decompress_workspace(uint64_t n_blocks, uint32_t block_size, int path, size_t max_bytes = ~(size_t)0);
/* fused one-wave-per-frame decoder (zstd_dec.hip): small batches, and the pipeline's irregular frames
 * (d_list != nullptr: decode blocks list_base + d_list[0 .. *d_list_n), n_blocks only sizes the grid) */
hipError_t launch_zstd_fused(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                             const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                             uint32_t block_size, uint64_t n_blocks, int32_t *d_status,
                             void *d_workspace, size_t workspace_bytes, const uint32_t *d_list,
                             const uint32_t *d_list_n, uint64_t list_base);
size_t zstd_fused_workspace(uint64_t n_blocks);

hipError_t launch_zstd_compress(hipStream_t s, const uint8_t *d_src, uint64_t src_stride, uint32_t block_size,
                                uint64_t n_blocks, uint8_t *d_dst, uint64_t dst_stride, int level,
                                uint32_t *d_out_size, int32_t *d_status, void *d_workspace, size_t workspace_bytes);
size_t zstd_compress_workspace(uint64_t n_blocks, int level, uint32_t block_size);
bool zstd_compress_supported(int level, uint32_t block_size);

} // namespace cryo

#define CRYO_WAVE 64
#define CRYO_ST_OK 0
#define CRYO_ST_CORRUPT (-4)

#endif
