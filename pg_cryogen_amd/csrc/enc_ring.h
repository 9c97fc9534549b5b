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

/* kStage: bytes per refill (64 lanes x 16 or 8 bytes) */
template <uint32_t kW, uint32_t kStage = kEncStage>
struct RingIn {
    static constexpr uint32_t kWM = kW - 1u;
    static constexpr uint32_t kPer = kStage / 64u;
    static_assert(kPer == 16u || kPer == 8u, "a refill is 16 or 8 bytes per lane");
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
        const uint32_t o = hi + lane * kPer;
        pre = make_uint4(0, 0, 0, 0);
        if (o + kPer <= n) __builtin_memcpy(&pre, src + o, kPer);
        else if (o < n) { /* the block's last, partial piece: never read past its end */
            uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
            for (uint32_t k = 0; k < kPer; k++) {
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
            if constexpr (kPer == 16u) *reinterpret_cast<uint4 *>(win + ((hi + lane * 16u) & kWM)) = pre;
            else *reinterpret_cast<uint2 *>(win + ((hi + lane * 8u) & kWM)) = make_uint2(pre.x, pre.y);
            hi += kStage;
            prefetch();
        }
    }
    /* the LDS behind the ring was used for something else: start staging again at the chunk that holds `pos` */
    __device__ inline void reopen(uint32_t pos)
    {
        hi = pos & ~(kStage - 1u);
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
    /* 4 bytes at any position of the block: ring if still there, global memory otherwise.  (The empty asm makes the loaded
     * value "used" inside the branch: the s_waitcnt vmcnt(0) a load from memory needs then stays in the branch instead of
     * sitting at the merge point, where every caller would wait for the output stores still on their way -- stores count on
     * vmcnt on gfx950.) */
    __device__ inline uint32_t rd32_any(uint32_t p) const
    {
        uint32_t v;
        if (p >= lo_pos()) v = rd32(p);
        else { __builtin_memcpy(&v, src + p, 4); asm volatile("" : "+v"(v)); }
        return v;
    }
    __device__ inline uint32_t byte_any(uint32_t p) const
    {
        uint32_t v;
        if (p >= lo_pos()) v = win[p & kWM];
        else { v = src[p]; asm volatile("" : "+v"(v)); }
        return v;
    }
    /* a byte the ring is known to hold */
    __device__ inline uint32_t byte_ring(uint32_t p) const { return win[p & kWM]; }
};

} // namespace
} // namespace cryo
