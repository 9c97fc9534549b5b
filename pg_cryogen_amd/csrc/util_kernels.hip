/*
 * util_kernels.hip -- generator, checksum and compare kernels (gfx950).
 *
 * All three are pure HBM streaming: every lane moves 16 bytes per access
 * (1 KiB per wave-instruction, the widest coalesced access on CDNA4) and the
 * grids are sized >> 256 workgroups so all 8 XCDs fill.
 */
#include "kernels.h"
#include "cryo_synth.h"

namespace cryo {

/* ---------------- synthetic blocks ---------------- */
__global__ void __launch_bounds__(256)
k_synth(uint64_t seed, uint64_t first_block, uint64_t block_step, uint64_t n_blocks, uint32_t B, int dist,
        uint8_t *__restrict__ dst, uint64_t dst_stride, uint32_t chunks_per_block)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t blk = gid / chunks_per_block;
    if (blk >= n_blocks) return;
    const uint32_t chunk = (uint32_t)(gid - blk * chunks_per_block);
    const uint32_t off0 = chunk * 16u;
    const cryo_synth_geom g = cryo_synth_geometry(B, dist);
    uint8_t *out = dst + blk * dst_stride;
    const uint64_t bi = first_block + blk * block_step;
    if (off0 + 16u <= B && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0)) {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++)
                v |= (uint32_t)cryo_synth_byte(seed, bi, B, dist, g, off0 + k * 4 + b) << (8 * b);
            w[k] = v;
        }
        *reinterpret_cast<uint4 *>(out + off0) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (uint32_t o = off0; o < B && o < off0 + 16u; o++)
            out[o] = cryo_synth_byte(seed, bi, B, dist, g, o);
    }
}

hipError_t launch_synth(hipStream_t s, uint64_t seed, uint64_t first_block, uint64_t block_step, uint64_t n_blocks,
                        uint32_t block_size, int dist, uint8_t *d_dst, uint64_t dst_stride)
{
    if (n_blocks == 0 || block_size == 0) return hipSuccess;
    const uint32_t cpb = (block_size + 15u) / 16u;
    const uint64_t threads = n_blocks * cpb;
    const uint64_t grid = (threads + 255) / 256;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_synth, dim3((uint32_t)grid), dim3(256), 0, s, seed, first_block, block_step, n_blocks,
                       block_size, dist, d_dst, dst_stride, cpb);
    return hipGetLastError();
}

/* ---------------- checksum ----------------
 * cryo_checksum64(p,n) = n*K0 + sum over 8-byte LE words w_i (tail zero padded)
 * of mix64(w_i + (i+1)*K1).  The sum is order independent, so a wave reduces
 * its block with 64 lanes striding over words.
 */
__device__ __host__ static inline uint64_t ck_mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256)
k_checksum(const uint8_t *__restrict__ src, uint64_t src_stride, const uint32_t *__restrict__ sizes,
           uint32_t fixed_size, uint64_t n_blocks, uint64_t *__restrict__ sums)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t blk = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (blk >= n_blocks) return;
    const uint8_t *p = src + blk * src_stride;
    const uint32_t n = sizes ? sizes[blk] : fixed_size;
    const uint32_t nwords = (n + 7u) >> 3;
    uint64_t acc = 0;
    const bool aligned = (reinterpret_cast<uintptr_t>(p) & 7u) == 0;
    for (uint32_t i = lane; i < nwords; i += 64u) {
        uint64_t w = 0;
        if (aligned && i * 8u + 8u <= n) {
            w = *reinterpret_cast<const uint64_t *>(p + (size_t)i * 8u);
        } else {
            for (uint32_t b = 0; b < 8u; b++) {
                uint32_t o = i * 8u + b;
                if (o < n) w |= (uint64_t)p[o] << (8u * b);
            }
        }
        acc += ck_mix(w + (uint64_t)(i + 1u) * 0x9E3779B97F4A7C15ull);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) sums[blk] = acc + (uint64_t)n * 0xD6E8FEB86659FD93ull;
}

hipError_t launch_checksum(hipStream_t s, const uint8_t *d_src, uint64_t src_stride,
                           const uint32_t *d_sizes, uint32_t fixed_size, uint64_t n_blocks,
                           uint64_t *d_sums)
{
    if (n_blocks == 0) return hipSuccess;
    const uint64_t grid = (n_blocks + 3) / 4;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_checksum, dim3((uint32_t)grid), dim3(256), 0, s, d_src, src_stride, d_sizes,
                       fixed_size, n_blocks, d_sums);
    return hipGetLastError();
}

/* ---------------- compare ---------------- */
__global__ void __launch_bounds__(256)
k_compare(const uint8_t *__restrict__ a, uint64_t a_stride, const uint8_t *__restrict__ b,
          uint64_t b_stride, uint32_t B, uint64_t n_blocks, uint64_t *__restrict__ mismatch)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t blk = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (blk >= n_blocks) return;
    const uint8_t *pa = a + blk * a_stride;
    const uint8_t *pb = b + blk * b_stride;
    int diff = 0;
    const bool aligned = ((reinterpret_cast<uintptr_t>(pa) | reinterpret_cast<uintptr_t>(pb)) & 15u) == 0;
    if (aligned) {
        const uint32_t nv = B >> 4;
        for (uint32_t i = lane; i < nv; i += 64u) {
            uint4 x = reinterpret_cast<const uint4 *>(pa)[i];
            uint4 y = reinterpret_cast<const uint4 *>(pb)[i];
            diff |= (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
        }
        for (uint32_t o = (nv << 4) + lane; o < B; o += 64u) diff |= pa[o] != pb[o];
    } else {
        for (uint32_t o = lane; o < B; o += 64u) diff |= pa[o] != pb[o];
    }
    const unsigned long long any = __ballot(diff);
    if (lane == 0 && any) atomicAdd((unsigned long long *)mismatch, 1ull);
}

hipError_t launch_compare(hipStream_t s, const uint8_t *d_a, uint64_t a_stride, const uint8_t *d_b,
                          uint64_t b_stride, uint32_t block_size, uint64_t n_blocks,
                          uint64_t *d_mismatch)
{
    if (n_blocks == 0) return hipSuccess;
    const uint64_t grid = (n_blocks + 3) / 4;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_compare, dim3((uint32_t)grid), dim3(256), 0, s, d_a, a_stride, d_b, b_stride,
                       block_size, n_blocks, d_mismatch);
    return hipGetLastError();
}

/* ---------------- gather: blocks from slots of the device-resident pool into a contiguous staging area ---------------- */
__global__ void __launch_bounds__(256)
k_gather_blocks(const uint8_t *__restrict__ pool, const GatherSlots slots, uint8_t *__restrict__ dst, uint32_t B, uint32_t first)
{
    /* one workgroup per 4 KiB piece of a block: 16 bytes per lane */
    const uint32_t pieces = (B + 4095u) >> 12;
    const uint32_t k = blockIdx.x / pieces, pc = blockIdx.x - k * pieces;
    const uint8_t *src = pool + (uint64_t)slots.slot[k] * B;
    uint8_t *out = dst + (uint64_t)(first + k) * B;
    const uint32_t o = (pc << 12) + threadIdx.x * 16u;
    if (o + 16u <= B && (B & 15u) == 0u) {
        *reinterpret_cast<uint4 *>(out + o) = *reinterpret_cast<const uint4 *>(src + o);
    } else {
        for (uint32_t i = o; i < o + 16u && i < B; i++) out[i] = src[i];
    }
}

hipError_t launch_gather_blocks(hipStream_t s, const uint8_t *d_pool, const GatherSlots &slots, uint32_t count, uint8_t *d_dst,
                                uint32_t block_size, uint32_t first)
{
    if (count == 0) return hipSuccess;
    const uint32_t pieces = (block_size + 4095u) >> 12;
    hipLaunchKernelGGL(k_gather_blocks, dim3(count * pieces), dim3(256), 0, s, d_pool, slots, d_dst, block_size, first);
    return hipGetLastError();
}

} // namespace cryo

/* host reference of the checksum (exported through the C ABI) */
extern "C" uint64_t cryo_checksum64(const void *p, size_t n)
{
    const uint8_t *b = static_cast<const uint8_t *>(p);
    const size_t nwords = (n + 7) >> 3;
    uint64_t acc = 0;
    for (size_t i = 0; i < nwords; i++) {
        uint64_t w = 0;
        for (size_t k = 0; k < 8; k++) {
            size_t o = i * 8 + k;
            if (o < n) w |= (uint64_t)b[o] << (8 * k);
        }
        acc += cryo::ck_mix(w + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull);
    }
    return acc + (uint64_t)n * 0xD6E8FEB86659FD93ull;
}
