/*
 * lz4_dec.hip -- LZ4 block decode, one wavefront (64 lanes) per cryo block.
 *
 * Replaces LZ4_decompress_safe(compressed, out, compressed_size, CRYO_BLCKSZ)
 * (reference compression.c:84) for a batch of independent blocks.
 *
 * Accept/reject rules are those of the LZ4 block format as liblz4 1.9.3's
 * bounds-checked decoder applies them (see oracle/lz4_oracle.c for the list);
 * additionally a block must decode to exactly block_size bytes.
 *
 * Structure of one wave: the sequence chain (token -> literal run -> offset ->
 * match) is inherently serial, so the parse is wave-uniform and the 64 lanes
 * co-operate on the two copies of each sequence:
 *   literals : lanes stride over the run, compressed stream -> output
 *   match    : lanes stride over the match, output -> output.  A match whose
 *              offset is smaller than its length (RLE-style, e.g. the zero gap
 *              of a cryo block: one ~110 KB match) is periodic with period
 *              `offset`, so every lane reads from the already-written prefix
 *              [op-off, op) and no lane depends on another lane of the same pass.
 * Vector memory operations of one wave are executed in order and the CU's L1
 * is coherent for its own stores, so a match may read bytes the same wave
 * stored in an earlier instruction without a fence.
 */
#include "kernels.h"

namespace cryo {

__device__ static inline uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__global__ void __launch_bounds__(256)
k_lz4_dec(const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
          const uint32_t *__restrict__ src_size, uint8_t *dst_base, uint64_t dst_stride, uint32_t B,
          uint64_t n_blocks, int32_t *__restrict__ status)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t blk = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (blk >= n_blocks) return;

    const uint8_t *__restrict__ src = src_base + src_off[blk];
    const uint32_t csize = src_size[blk];
    uint8_t *dst = dst_base + blk * dst_stride;

    uint32_t ip = 0, op = 0;
    bool bad = (csize == 0);

    while (!bad) {
        const uint32_t token = uni(src[ip]);
        ip++;
        uint32_t len = token >> 4;
        if (len == 15u) {
            if (ip + 15u >= csize) { bad = true; break; }
            uint32_t s;
            do {
                s = uni(src[ip]);
                ip++;
                len += s;
                if (ip + 15u >= csize) break;
            } while (s == 255u);
        }
        /* literal run: must be the final one if it comes near either end */
        const bool last = (op + len + 12u > B) || (ip + len + 8u > csize);
        if (last && (ip + len != csize || op + len > B)) { bad = true; break; }
        for (uint32_t i = lane; i < len; i += 64u) dst[op + i] = src[ip + i];
        ip += len;
        op += len;
        if (last) break;

        const uint32_t off = uni((uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8));
        ip += 2;
        len = token & 15u;
        if (len == 15u) {
            uint32_t s;
            do {
                s = uni(src[ip]);
                ip++;
                len += s;
                if (ip + 4u >= csize) { bad = true; break; }
            } while (s == 255u);
            if (bad) break;
        }
        len += 4u;
        if (off > op || op + len + 5u > B) { bad = true; break; }

        if (off == 0u) {
            /* liblz4 1.9.3 does not reject offset 0; it reproduces as zero bytes */
            for (uint32_t i = lane; i < len; i += 64u) dst[op + i] = 0;
        } else if (off >= 64u || off >= len) {
            /* each pass reads only bytes written by earlier passes or earlier sequences */
            for (uint32_t i = lane; i < len; i += 64u) dst[op + i] = dst[op + i - off];
        } else {
            /* overlapping match: periodic with period off, source is the prefix [op-off, op) */
            const uint8_t *pat = dst + op - off;
            uint32_t ph = lane % off;
            const uint32_t adv = 64u % off;
            for (uint32_t i = lane; i < len; i += 64u) {
                dst[op + i] = pat[ph];
                ph += adv;
                if (ph >= off) ph -= off;
            }
        }
        op += len;
    }
    if (!bad && op != B) bad = true;
    if (lane == 0) status[blk] = bad ? CRYO_ST_CORRUPT : CRYO_ST_OK;
}

hipError_t launch_lz4_decompress(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                                 const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                                 uint32_t block_size, uint64_t n_blocks, int32_t *d_status)
{
    if (n_blocks == 0) return hipSuccess;
    const uint64_t grid = (n_blocks + 3) / 4;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_lz4_dec, dim3((uint32_t)grid), dim3(256), 0, s, d_src, d_src_off,
                       d_src_size, d_dst, dst_stride, block_size, n_blocks, d_status);
    return hipGetLastError();
}

} // namespace cryo
