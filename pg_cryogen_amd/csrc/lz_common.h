/*
 * lz_common.h -- wave-level building blocks shared by the LZ4 and zstd block decoders.
 *
 *   Wave<R>: the per-wave data movement both decoders use --
 *     HBM --16 B/lane--> 2 KiB input ring in LDS --one byte per lane--> 64-byte window (VGPRs)
 *     window / output ring --> output ring (LDS, last R bytes) --16 B/lane--> HBM
 *   plus cross-lane helpers (readlane, DPP scans) and the wave-cooperative copies of one
 *   LZ sequence (literal run from the staged byte stream, match from the ring / the output).
 */
#ifndef CRYO_LZ_COMMON_H
#define CRYO_LZ_COMMON_H

#include "kernels.h"

namespace cryo {
namespace {


constexpr uint32_t kInRing = 2048, kInMask = kInRing - 1; /* >= kWMax + 256 + 72 + kChunk */
constexpr uint32_t kChunk = 1024;   /* bytes per output flush: 64 lanes x 16 B */
constexpr uint32_t kInChunk = 512;  /* bytes per input refill: 64 lanes x 8 B (leaves room for a 1 KiB parse window) */

__device__ inline uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
/* NOTE: never rebuild a pointer from integers (it becomes a FLAT pointer and every access then
 * also counts on lgkmcnt, serialising LDS traffic behind HBM traffic); make the OFFSET uniform. */
__device__ inline uint64_t uni64(uint64_t v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ inline uint32_t lane_get(uint32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, l); }
__device__ inline uint32_t ctz64(unsigned long long m) { return (uint32_t)__builtin_ctzll(m); }

template <uint32_t R>
struct Wave {
    /* LDS */
    uint8_t *ring; /* R bytes   */
    uint8_t *in;   /* kInRing   */
    /* stream */
    const uint8_t *abase; /* 16-byte aligned address at or below the block's first byte */
    uint32_t delta;       /* first byte = abase + delta                                */
    uint32_t vend;        /* delta + csize: end of stream in "virtual" positions       */
    uint32_t in_hi;       /* virtual position staged up to (multiple of kInChunk)      */
    uint2 pre;            /* prefetched next chunk                                     */
    /* output */
    uint8_t *dst;
    uint32_t op;      /* bytes produced      */
    uint32_t flushed; /* bytes stored to HBM (multiple of kChunk) */
    bool dst_aligned;
    uint32_t lane;

    __device__ inline void prefetch()
    {
        const uint32_t o = in_hi + lane * 8u;
        pre = make_uint2(0, 0);
        if (o < vend) pre = *reinterpret_cast<const uint2 *>(abase + o);
    }
    /* write the prefetched chunk into the input ring, start fetching the one after */
    __device__ inline void refill()
    {
        *reinterpret_cast<uint2 *>(in + ((in_hi + lane * 8u) & kInMask)) = pre;
        in_hi += kInChunk;
        prefetch();
    }
    /* keep at least 128 staged bytes ahead of virtual position vp (the chunk after that is
     * already on its way in `pre`) */
    __device__ inline void need(uint32_t vp)
    {
        while (in_hi < vend && vp + 128u > in_hi) refill();
    }
    /* 64-byte window at virtual position vp: lane l holds byte vp + l */
    __device__ inline uint32_t window(uint32_t vp) const { return in[(vp + lane) & kInMask]; }

    /* store completed 1 KiB output chunks */
    __device__ inline void flush()
    {
        while (op - flushed >= kChunk) {
            if (dst_aligned) {
                const uint4 x = *reinterpret_cast<const uint4 *>(ring + ((flushed + lane * 16u) & (R - 1)));
                *reinterpret_cast<uint4 *>(dst + flushed + lane * 16u) = x;
            } else {
                for (uint32_t i = lane; i < kChunk; i += 64u) dst[flushed + i] = ring[(flushed + i) & (R - 1)];
            }
            flushed += kChunk;
        }
    }
    __device__ inline void flush_tail()
    {
        for (uint32_t i = flushed + lane; i < op; i += 64u) dst[i] = ring[i & (R - 1)];
        flushed = op;
    }
};

/* wave64 inclusive add-scan on the DPP cross-lane network (no LDS round trips):
 * Hillis-Steele inside each row of 16 (row_shr 1,2,4,8; out-of-row sources read 0), then
 * row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3. */
template <int CTRL, int ROW_MASK>
__device__ inline uint32_t dpp_add(uint32_t x)
{
    return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}
__device__ inline uint32_t scan16_incl(uint32_t x) /* independent scan in every row of 16 lanes */
{
    x = dpp_add<0x111, 0xf>(x);
    x = dpp_add<0x112, 0xf>(x);
    x = dpp_add<0x114, 0xf>(x);
    x = dpp_add<0x118, 0xf>(x);
    return x;
}
__device__ inline uint32_t scan64_incl(uint32_t x)
{
    x = scan16_incl(x);
    x = dpp_add<0x142, 0xa>(x);
    x = dpp_add<0x143, 0xc>(x);
    return x;
}

/* lane % m for lane < 64, 0 < m < 64 */
__device__ inline uint32_t lane_mod(uint32_t lane, uint32_t m)
{
    const uint32_t q = (uint32_t)((float)lane * __frcp_rn((float)m));
    int32_t r = (int32_t)(lane - q * m);
    if (r < 0) r += (int32_t)m;
    if (r >= (int32_t)m) r -= (int32_t)m;
    return (uint32_t)r;
}


/*
 * Copy `ll` literal bytes that start at virtual position p of the staged stream into the
 * output ring (64 bytes per step).  Returns the position after the run.
 */
template <uint32_t R>
__device__ inline uint32_t wave_copy_literals(Wave<R> &w, uint32_t p, uint32_t ll)
{
    uint32_t rem = ll;
    while (rem) {
        w.flush();
        w.need(p);
        const uint32_t x = w.window(p);
        const uint32_t n = rem < 64u ? rem : 64u;
        if (w.lane < n) w.ring[(w.op + w.lane) & (R - 1)] = (uint8_t)x;
        w.op += n;
        p += n;
        rem -= n;
    }
    return p;
}

/*
 * Copy a match of `ml` bytes at distance `off` (1 <= off <= w.op).  Near sources come from
 * the LDS ring (overlapping matches are periodic: first 64 bytes via lane % off, later pieces
 * via the smallest multiple of the period >= 64); far sources (off > R - 128) were already
 * flushed and are read back from the output buffer.  off == 0 writes zero bytes (what
 * liblz4 1.9.3 does; zstd never calls it with 0).
 */
template <uint32_t R>
__device__ inline void wave_copy_match(Wave<R> &w, uint32_t off, uint32_t ml)
{
    const uint32_t lane = w.lane;
    uint32_t rem = ml;
    if (off == 0u) {
        while (rem) {
            w.flush();
            const uint32_t n = rem < 64u ? rem : 64u;
            if (lane < n) w.ring[(w.op + lane) & (R - 1)] = 0;
            w.op += n;
            rem -= n;
        }
    } else if (off <= R - 128u) {
        uint32_t eff = off; /* distance used by pieces after the first */
        uint32_t sidx;      /* ring index this lane reads for the first piece */
        if (off < 64u && ml > off) {
            sidx = w.op - off + lane_mod(lane, off);
            /* out[i] = out[i - eff] holds for every i >= op + 64 only while eff <= 64 + off */
            eff = (uint32_t)(64.0f * __frcp_rn((float)off)) * off;
            if (eff >= 64u + off) eff -= off;
            if (eff < 64u) eff += off;
        } else {
            sidx = w.op - off + lane;
        }
        {
            const uint32_t n = rem < 64u ? rem : 64u;
            const uint8_t x = w.ring[sidx & (R - 1)];
            if (lane < n) w.ring[(w.op + lane) & (R - 1)] = x;
            w.op += n;
            rem -= n;
        }
        while (rem) {
            w.flush();
            const uint32_t n = rem < 64u ? rem : 64u;
            const uint8_t x = w.ring[(w.op - eff + lane) & (R - 1)];
            if (lane < n) w.ring[(w.op + lane) & (R - 1)] = x;
            w.op += n;
            rem -= n;
        }
    } else {
        while (rem) {
            w.flush();
            const uint32_t n = rem < 64u ? rem : 64u;
            uint8_t x = 0;
            if (lane < n) x = w.dst[w.op - off + lane];
            if (lane < n) w.ring[(w.op + lane) & (R - 1)] = x;
            w.op += n;
            rem -= n;
        }
    }
}

} // namespace
} // namespace cryo

#endif
