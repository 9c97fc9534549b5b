/*
 * enc_ring.h -- input side shared by the batch encoders (lz4_enc2.hip, zstd_enc.hip): the most recent
 * kW bytes of a block in an LDS ring, staged 1 KiB at a time with the next chunk's global load already
 * in flight.  Positions are byte offsets from the block's first byte; a position older than the ring is
 * read from global memory (normally an L2 hit: the block has just streamed through).
 */
#pragma once
#include "lz_common.h"

namespace cryo {
namespace {

constexpr uint32_t kEncStage = 1024;

template <uint32_t kW>
struct RingIn {
    static constexpr uint32_t kWM = kW - 1u;
    uint8_t *win; /* LDS, kW bytes, 16-byte aligned */
    const uint8_t *src;
    uint32_t n, hi, lane;
    uint32_t floor; /* lowest position the ring may hold (after reopen(): what was staged before is gone) */
    uint4 pre;

    __device__ inline void open(uint8_t *lds, const uint8_t *s, uint32_t len, uint32_t ln)
    {
        win = lds; src = s; n = len; hi = 0; lane = ln; floor = 0;
        prefetch();
    }
    __device__ inline void prefetch()
    {
        const uint32_t o = hi + lane * 16u;
        pre = make_uint4(0, 0, 0, 0);
        if (o + 16u <= n) __builtin_memcpy(&pre, src + o, 16);
        else if (o < n) { /* the block's last, partial 16 bytes: never read past its end */
            uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
            for (uint32_t k = 0; k < 16u; k++) {
                const uint32_t b = o + k < n ? (uint32_t)src[o + k] << (8u * (k & 3u)) : 0u;
                if (k < 4u) w0 |= b; else if (k < 8u) w1 |= b; else if (k < 12u) w2 |= b; else w3 |= b;
            }
            pre = make_uint4(w0, w1, w2, w3);
        }
    }
    /* stage until position `upto` (exclusive) is in the ring */
    __device__ inline void ensure(uint32_t upto)
    {
        while (hi < n && hi < upto) {
            *reinterpret_cast<uint4 *>(win + ((hi + lane * 16u) & kWM)) = pre;
            hi += kEncStage;
            prefetch();
        }
    }
    /* the LDS behind the ring was used for something else: start staging again at the chunk that holds `pos` */
    __device__ inline void reopen(uint32_t pos)
    {
        hi = pos & ~(kEncStage - 1u);
        floor = hi;
        prefetch();
    }
    __device__ inline uint32_t lo_pos() const
    {
        const uint32_t l = hi > kW ? hi - kW : 0u;
        return l > floor ? l : floor;
    }

    /* dword i (0, 1, 2) of the aligned group that holds bytes p .. p+11 */
    __device__ inline uint32_t dw(uint32_t p, uint32_t i) const
    {
        return reinterpret_cast<const uint32_t *>(win)[((p >> 2) + i) & (kW / 4 - 1)];
    }
    __device__ inline uint32_t rd32(uint32_t p) const { return __builtin_amdgcn_alignbyte(dw(p, 1), dw(p, 0), p & 3u); }
    /* 4 bytes at any position of the block: ring if still there, global memory otherwise */
    __device__ inline uint32_t rd32_any(uint32_t p) const
    {
        if (p >= lo_pos()) return rd32(p);
        uint32_t v;
        __builtin_memcpy(&v, src + p, 4);
        return v;
    }
    __device__ inline uint32_t byte_any(uint32_t p) const { return p >= lo_pos() ? (uint32_t)win[p & kWM] : (uint32_t)src[p]; }
};

} // namespace
} // namespace cryo
