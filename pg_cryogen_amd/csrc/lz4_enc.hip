/*
 * lz4_enc.hip -- LZ4 block encode, one wavefront per cryo block, output bytes
 * identical to liblz4 1.9.3.
 *
 * Replaces LZ4_compress_fast(data, out, CRYO_BLCKSZ, LZ4_compressBound(CRYO_BLCKSZ),
 * lz4_acceleration_guc) (reference compression.c:70-72).
 *
 * The greedy parser is a serial recurrence over one hash table (every probe
 * reads the slot the previous probe wrote), so to stay bit-exact the probe
 * chain runs wave-uniform; the position table (16 KiB: 4096 x u32, or
 * 8192 x u16 for inputs below 65547 bytes) lives in LDS, one per wave.
 * The 64 lanes co-operate on the parts that are data parallel:
 *   - zeroing the table,
 *   - forward match extension (64 bytes compared per step, ballot + ctz),
 *   - literal copies into the output.
 */
#include "kernels.h"
#include <cstdlib>

namespace cryo {

namespace {

constexpr uint32_t kMinMatch = 4, kMfLimit = 12, kLastLiterals = 5, kMinLength = 13;
constexpr uint32_t kMaxDist = 65535, kSkipTrigger = 6, kLimit64k = 65536 + 11;

__device__ inline uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ inline uint32_t ld32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* hash of the 4 (small table) or 5 (large table) bytes at p */
__device__ inline uint32_t hash_at(const uint8_t *p, bool small)
{
    const uint32_t lo = ld32(p);
    if (small) return (lo * 2654435761u) >> (32 - 13);
    const uint64_t v = (uint64_t)lo | ((uint64_t)p[4] << 32);
    return (uint32_t)(((v << 24) * 889523592379ull) >> (64 - 12));
}

__device__ inline uint32_t tab_get(const uint32_t *tab, uint32_t h, bool small)
{
    return small ? reinterpret_cast<const uint16_t *>(tab)[h] : tab[h];
}
__device__ inline void tab_put(uint32_t *tab, uint32_t h, uint32_t v, bool small)
{
    if (small) reinterpret_cast<uint16_t *>(tab)[h] = (uint16_t)v;
    else tab[h] = v;
}

/* 255-run length code, written by lane 0; returns the new output position */
__device__ inline uint32_t put_len(uint8_t *dst, uint32_t o, uint32_t len, uint32_t lane)
{
    const uint32_t n255 = len / 255u;
    for (uint32_t i = lane; i < n255; i += 64u) dst[o + i] = 255;
    if (lane == 0) dst[o + n255] = (uint8_t)(len - n255 * 255u);
    return o + n255 + 1u;
}

} // namespace

__global__ void __launch_bounds__(256)
k_lz4_enc(const uint8_t *__restrict__ src_base, uint64_t src_stride, uint32_t n, uint64_t n_blocks,
          uint8_t *__restrict__ dst_base, uint64_t dst_stride, int accel_in,
          uint32_t *__restrict__ out_size, int32_t *__restrict__ status)
{
    __shared__ uint32_t tables[4][4096];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = threadIdx.x >> 6;
    const uint64_t blk = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid;
    if (blk >= n_blocks) return;

    const uint8_t *__restrict__ src = src_base + blk * src_stride;
    uint8_t *__restrict__ dst = dst_base + blk * dst_stride;
    uint32_t *tab = tables[wid];
    const bool small = n < kLimit64k;
    uint32_t accel = accel_in < 1 ? 1u : (accel_in > 65537 ? 65537u : (uint32_t)accel_in);

    for (uint32_t i = lane; i < 4096u; i += 64u) tab[i] = 0;
    __builtin_amdgcn_wave_barrier();

    uint32_t ip = 0, anchor = 0, op = 0;

    if (n >= kMinLength) {
        const uint32_t mflimit_p1 = n - kMfLimit + 1u;
        const uint32_t matchlimit = n - kLastLiterals;

        tab_put(tab, uni(hash_at(src, small)), 0, small);
        ip = 1;
        uint32_t fwd_h = uni(hash_at(src + ip, small));
        bool done = false;

        while (!done) {
            uint32_t match = 0;
            /* ---- probe chain (wave-uniform) ---- */
            {
                uint32_t fwd = ip, step = 1, nb = accel << kSkipTrigger;
                for (;;) {
                    const uint32_t h = fwd_h;
                    const uint32_t cur = fwd;
                    ip = fwd;
                    fwd += step;
                    step = nb++ >> kSkipTrigger;
                    if (fwd > mflimit_p1) { done = true; break; }
                    match = uni(tab_get(tab, h, small));
                    fwd_h = uni(hash_at(src + fwd, small));
                    tab_put(tab, h, cur, small);
                    if (!small && match + kMaxDist < cur) continue;
                    if (uni(ld32(src + match)) == uni(ld32(src + ip))) break;
                }
            }
            if (done) break;
            /* ---- extend backwards ---- */
            while (ip > anchor && match > 0 && uni(src[ip - 1]) == uni(src[match - 1])) { ip--; match--; }

            /* ---- literal run ---- */
            uint32_t tok = op++;
            uint32_t tokval;
            {
                const uint32_t lit = ip - anchor;
                if (lit >= 15u) { tokval = 15u << 4; op = put_len(dst, op, lit - 15u, lane); }
                else tokval = lit << 4;
                for (uint32_t i = lane; i < lit; i += 64u) dst[op + i] = src[anchor + i];
                op += lit;
            }
            for (;;) {
                /* ---- offset, then forward extension: 64 bytes per step ---- */
                if (lane == 0) {
                    dst[op] = (uint8_t)(ip - match);
                    dst[op + 1] = (uint8_t)((ip - match) >> 8);
                }
                op += 2;
                uint32_t a = ip + kMinMatch, b = match + kMinMatch;
                for (;;) {
                    const bool inb = a + lane < matchlimit;
                    const bool eq = inb && src[a + lane] == src[b + lane];
                    const unsigned long long neq = __ballot(!eq);
                    if (neq != 0ull) { a += (uint32_t)__builtin_ctzll(neq); break; }
                    a += 64u; b += 64u;
                }
                const uint32_t ml = a - (ip + kMinMatch);
                ip = a;
                if (ml >= 15u) { tokval += 15u; op = put_len(dst, op, ml - 15u, lane); }
                else tokval += ml;
                if (lane == 0) dst[tok] = (uint8_t)tokval;
                anchor = ip;
                if (ip >= mflimit_p1) { done = true; break; }

                tab_put(tab, uni(hash_at(src + ip - 2, small)), ip - 2u, small);
                /* ---- immediate re-test at ip ---- */
                const uint32_t h = uni(hash_at(src + ip, small));
                match = uni(tab_get(tab, h, small));
                tab_put(tab, h, ip, small);
                if ((small || match + kMaxDist >= ip) && uni(ld32(src + match)) == uni(ld32(src + ip))) {
                    tok = op++;
                    tokval = 0;
                    continue;
                }
                break;
            }
            if (done) break;
            fwd_h = uni(hash_at(src + (++ip), small));
        }
    }
    /* ---- last literals ---- */
    {
        const uint32_t lit = n - anchor;
        const uint32_t tok = op++;
        if (lit >= 15u) { if (lane == 0) dst[tok] = 15u << 4; op = put_len(dst, op, lit - 15u, lane); }
        else if (lane == 0) dst[tok] = (uint8_t)(lit << 4);
        for (uint32_t i = lane; i < lit; i += 64u) dst[op + i] = src[anchor + i];
        op += lit;
    }
    if (lane == 0) { out_size[blk] = op; status[blk] = CRYO_ST_OK; }
}

hipError_t launch_lz4_compress(hipStream_t s, const uint8_t *d_src, uint64_t src_stride,
                               uint32_t block_size, uint64_t n_blocks, uint8_t *d_dst,
                               uint64_t dst_stride, int accel, uint32_t *d_out_size,
                               int32_t *d_status)
{
    if (n_blocks == 0) return hipSuccess;
    /* liblz4's byU32 table mode, positions below 2^24: the 64-probes-per-step kernel (lz4_enc2.hip) */
    static const bool serial_only = cryo_tuning_env("CRYO_LZ4_ENC") && cryo_tuning_env("CRYO_LZ4_ENC")[0] == '1'; /* testing aid */
    if (!serial_only && block_size >= kLimit64k && block_size <= (16u << 20))
        return launch_lz4_compress_batch64(s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride, accel,
                                           d_out_size, d_status);
    const uint64_t grid = (n_blocks + 3) / 4;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_lz4_enc, dim3((uint32_t)grid), dim3(256), 0, s, d_src, src_stride,
                       block_size, n_blocks, d_dst, dst_stride, accel, d_out_size, d_status);
    return hipGetLastError();
}

} // namespace cryo
