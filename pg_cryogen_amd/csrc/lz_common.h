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
/* votes: HIP's __ballot / __any turn the predicate into 0/1 in a register and compare that again (v_cndmask + v_cmp per vote);
 * the builtin takes the compare's lane mask as it stands.  The decoders are bound by vector instruction issue at full
 * occupancy (k_lz4_dec_seq: 88 % of the SIMDs' cycles, NOTEBOOK.md 4.1), and a batch holds some twenty votes. */
__device__ inline unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ inline bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ inline bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }
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

/* 16 bytes per lane of decoded output on their way to memory, for the LONG runs (a literal run or a pattern streamed 1 KiB
 * per instruction for tens of KiB: the zero gap of a cryo block).  A wave stores into its own block: with
 * plain stores 65 536 such streams reach 3.5 TB/s, with non-temporal ones 5.65 (profiles/r05_store_bw.txt; a grid-stride fill
 * reaches 5.0-5.7 either way, hipMemsetAsync 6.3) -- `zeros` 3 287 -> 5 258 GB/s, `narrow` 2 077 -> 3 191, `int4` 2 424 -> 3 537
 * (profiles/r05_nt_ab.txt).  The line stays in the XCD's L2 all the same (MI355X_MICROARCH.md, "stores of each flavour").
 * The batches' 1 KiB flushes stay plain stores: non-temporal ones there made the headline batch 1-4 % slower (same file). */
#ifndef CRYO_NT_STREAM
#define CRYO_NT_STREAM 1
#endif
#ifndef CRYO_NT_FLUSH
#define CRYO_NT_FLUSH 0
#endif
template <bool NT>
__device__ inline void store16_out(uint8_t *p, const uint4 v)
{
    if constexpr (NT) {
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        const u32x4_ y = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(y, reinterpret_cast<u32x4_ *>(p));
    } else {
        *reinterpret_cast<uint4 *>(p) = v;
    }
}

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
    uint2 pre, pre2, pre3; /* the next three chunks, on their way (a batch consumes up to three) */
    uint32_t nstale;       /* trailing prefetch slots emptied by refill_deferred() and not requested again yet */
    /* output */
    uint8_t *dst;
    uint32_t op;      /* bytes produced      */
    uint32_t flushed; /* bytes stored to HBM (multiple of kChunk) */
    bool dst_aligned;
    uint32_t lane;

    /* A SEGMENTED stream (zstd decode, round 6): the stream's bytes lie in up to 64 pieces of a scratch area, in order --
     * what the Huffman walkers of k_zhufw leave: piece j = s_src bytes into sbase, the stream's positions [s_end - length,
     * s_end); lane j holds piece j.  Reading them in place takes the copy into a contiguous pool (k_zmove: 3 GB read and
     * 5.4 GB written per call of 65 536 frames) out of the pipeline.  Every piece that is not empty has at least 8 bytes
     * (the producer sees to it), and up to 15 bytes behind a piece's last may be read. */
    bool seg_on = false;
    const uint8_t *sbase = nullptr;
    uint32_t s_src = 0, s_end = 0, s_dst = 0, s_nxt = 0; /* per lane: piece j; s_nxt: s_src of the next piece that is not empty */
    uint32_t cs = 0;                                     /* first piece that ends behind the last fetch's position */

    __device__ inline uint2 fetch_seg(uint32_t at)
    {
        const uint32_t o = at + lane * 8u;
        uint2 v = make_uint2(0, 0);
        if (at >= vend) return v;
        while (cs < 63u && (uint32_t)__builtin_amdgcn_readlane((int)s_end, (int)cs) <= at) cs++;
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)s_end, (int)cs);
        uint32_t so, eseg, nx;
        if (e0 >= at + kInChunk || cs >= 63u) { /* the whole chunk lies in one piece */
            so = (uint32_t)__builtin_amdgcn_readlane((int)s_src, (int)cs) + (o - (uint32_t)__builtin_amdgcn_readlane((int)s_dst, (int)cs));
            eseg = e0;
            nx = (uint32_t)__builtin_amdgcn_readlane((int)s_nxt, (int)cs);
        } else {
            uint32_t seg = cs;
            for (uint32_t k = cs; k < 63u; k++) {
                const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)s_end, (int)k);
                if (e >= at + kInChunk) break;
                seg += o >= e ? 1u : 0u;
            }
            const int a = (int)(seg << 2);
            so = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)s_src) + (o - (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)s_dst));
            eseg = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)s_end);
            nx = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)s_nxt);
        }
        if (o < vend) {
            __builtin_memcpy(&v, sbase + so, 8);
            if (o + 8u > eseg && eseg < vend) { /* the piece ends inside these 8 bytes: the rest is the head of the next one */
                uint2 t;
                __builtin_memcpy(&t, sbase + nx, 8);
                const uint32_t r = 8u * (eseg - o); /* bits that are this piece's */
                const uint64_t a64 = ((uint64_t)v.y << 32) | v.x, b64 = ((uint64_t)t.y << 32) | t.x;
                const uint64_t m = (a64 & ((1ull << r) - 1ull)) | (b64 << r);
                v = make_uint2((uint32_t)m, (uint32_t)(m >> 32));
            }
        }
        return v;
    }
    __device__ inline uint2 fetch(uint32_t at)
    {
        if (seg_on) return fetch_seg(at);
        const uint32_t o = at + lane * 8u;
        uint2 v = make_uint2(0, 0);
        if (o < vend) v = *reinterpret_cast<const uint2 *>(abase + o);
        return v;
    }
    __device__ inline void prefetch()
    {
        pre = fetch(in_hi);
        pre2 = fetch(in_hi + kInChunk);
        pre3 = fetch(in_hi + 2u * kInChunk);
        nstale = 0;
    }
    /* write the oldest prefetched chunk into the input ring, start fetching the one after the others */
    __device__ inline void refill()
    {
        *reinterpret_cast<uint2 *>(in + ((in_hi + lane * 8u) & kInMask)) = pre;
        in_hi += kInChunk;
        pre = pre2;
        pre2 = pre3;
        pre3 = fetch(in_hi + 2u * kInChunk);
    }
    /* the same without the new request (at most three times between two top_up() calls): a wave whose next
     * wait on memory must not include a load issued a moment ago (vmcnt counts in order) stages at the start of
     * a batch and requests in the middle of it */
    __device__ inline void refill_deferred()
    {
        *reinterpret_cast<uint2 *>(in + ((in_hi + lane * 8u) & kInMask)) = pre;
        in_hi += kInChunk;
        pre = pre2;
        pre2 = pre3;
        nstale++;
    }
    __device__ inline void top_up()
    {
        if (nstale >= 3u) pre = fetch(in_hi);
        if (nstale >= 2u) pre2 = fetch(in_hi + kInChunk);
        if (nstale >= 1u) pre3 = fetch(in_hi + 2u * kInChunk);
        nstale = 0;
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
                store16_out<CRYO_NT_FLUSH != 0>(dst + flushed + lane * 16u, x);
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
 * Long literal runs (incompressible blocks are one literal run).  At a 1 KiB boundary with everything
 * flushed, whole chunks go from the input stream to the output 16 bytes per lane, global to global; only the
 * last R bytes also pass through the output ring (later matches may reach them there).  The input ring is
 * re-synchronised behind the run.  Returns false when the conditions do not hold (yet).
 */
template <uint32_t R>
__device__ inline bool wave_stream_literals(Wave<R> &w, uint32_t &p, uint32_t &rem)
{
    if (rem < 2u * R || w.flushed != w.op || !w.dst_aligned || w.seg_on /* pieces: through the ring */) return false;
    const uint32_t nch = rem / kChunk;
    for (uint32_t c = 0; c < nch; c++) {
        uint4 v;
        __builtin_memcpy(&v, w.abase + p + c * kChunk + w.lane * 16u, 16);
        store16_out<CRYO_NT_STREAM != 0>(w.dst + w.op + c * kChunk + w.lane * 16u, v);
        if (c + R / kChunk >= nch) *reinterpret_cast<uint4 *>(w.ring + ((w.op + c * kChunk + w.lane * 16u) & (R - 1))) = v;
    }
    w.op += nch * kChunk;
    w.flushed = w.op;
    p += nch * kChunk;
    rem -= nch * kChunk;
    w.in_hi = p & ~(kInChunk - 1u); /* nothing before p is looked at again */
    w.prefetch();
    return true;
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
        if (wave_stream_literals(w, p, rem)) continue;
        w.need(p);
        const uint32_t x = w.window(p);
        uint32_t n = rem < 64u ? rem : 64u;
        if (rem >= 3u * R) { const uint32_t to = kChunk - (w.op & (kChunk - 1u)); n = n < to ? n : to; } /* land on the 1 KiB boundary */
        if (w.lane < n) w.ring[(w.op + w.lane) & (R - 1)] = (uint8_t)x;
        w.op += n;
        p += n;
        rem -= n;
    }
    return p;
}

/*
 * Long runs.  At a 1 KiB boundary with everything flushed, when the output continues a pattern whose
 * period divides 16 (run of one byte, 2/4/8/16-byte records -- the zero gap of a cryo block is the common
 * case) or is all zero (`zero`), the 16 bytes before op are the pattern for every aligned
 * 16-byte slot to come: fill the ring with it once and stream whole 1 KiB chunks straight to the output,
 * 16 bytes per lane and one store per KiB instead of 16 byte-wise LDS round trips.  The ring stays valid
 * (every position holds pattern[x mod 16]).  Returns false when the conditions do not hold (yet).
 */
template <uint32_t R>
__device__ inline bool wave_stream_pattern(Wave<R> &w, uint32_t &rem, bool zero)
{
    if (rem < 2u * R || w.flushed != w.op || !w.dst_aligned || (!zero && w.op < 16u)) return false;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!zero) v = *reinterpret_cast<const uint4 *>(w.ring + ((w.op - 16u) & (R - 1)));
#pragma unroll
    for (uint32_t k = 0; k < R / kChunk; k++) *reinterpret_cast<uint4 *>(w.ring + ((k * kChunk + w.lane * 16u) & (R - 1))) = v;
    const uint32_t nch = rem / kChunk;
    for (uint32_t c = 0; c < nch; c++) store16_out<CRYO_NT_STREAM != 0>(w.dst + w.op + c * kChunk + w.lane * 16u, v);
    w.op += nch * kChunk;
    w.flushed = w.op;
    rem -= nch * kChunk;
    return true;
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
            if (wave_stream_pattern(w, rem, true)) continue;
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
        const bool pow2_period = off <= 16u && (off & (off - 1u)) == 0u; /* the period divides 16 */
        while (rem) {
            w.flush();
            if (pow2_period && wave_stream_pattern(w, rem, false)) continue;
            uint32_t n = rem < 64u ? rem : 64u;
            if (pow2_period && rem >= 3u * R) { const uint32_t to = kChunk - (w.op & (kChunk - 1u)); n = n < to ? n : to; } /* land on the 1 KiB boundary */
            const uint8_t x = w.ring[(w.op - eff + lane) & (R - 1)];
            if (lane < n) w.ring[(w.op + lane) & (R - 1)] = x;
            w.op += n;
            rem -= n;
        }
    } else {
        while (rem) {
            w.flush();
            if (rem >= kChunk && w.flushed == w.op && w.dst_aligned && off >= kChunk) {
                /* a whole flushed 1 KiB chunk from the output buffer: 16 bytes per lane, ring kept current */
                uint4 v;
                __builtin_memcpy(&v, w.dst + w.op - off + lane * 16u, 16);
                *reinterpret_cast<uint4 *>(w.ring + ((w.op + lane * 16u) & (R - 1))) = v;
                w.op += kChunk;
                rem -= kChunk;
                continue;
            }
            const uint32_t n = rem < 64u ? rem : 64u;
            uint8_t x = 0;
            if (lane < n) x = w.dst[w.op - off + lane];
            if (lane < n) w.ring[(w.op + lane) & (R - 1)] = x;
            w.op += n;
            rem -= n;
        }
    }
}


/* (the copy engine of the block decoders lives in lz4_copy.h) */
constexpr uint32_t kTMax = 1024;  /* output bytes per batch (16 chunks of 64) */
constexpr uint32_t kNCh = kTMax / 64;

struct Stats {
    uint32_t batches, batch_seqs, general_seqs, chunks, rounds, zero_batches;
    uint32_t ablate;
    unsigned long long t[8]; /* cycle stamps per phase (diagnostic build only) */
    unsigned long long t0;
    bool on;
};
/* phase stamp: adds the cycles since the previous stamp to bucket k (STATS build only) */
__device__ inline void stamp(Stats &st, int k)
{
    if (st.on) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        st.t[k] += now - st.t0;
        st.t0 = now;
    }
}


} // namespace
/* ---- helpers of the encoders that work against global memory (zstd_dfast.h) ---- */
/* per-lane unaligned loads */
__device__ inline uint64_t ld64v(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ inline uint32_t ld32v(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

/* Continuation of a forward match count from `done` bytes on: long matches (the zero gap of a cryo block is one match of
 * ~110 KB) are compared 16 bytes per lane, two KiB per trip to memory, instead of one byte per lane (round 6:
 * `zeros` 527 -> see profiles/r06_zstd_enc.txt); the last stretch in front of `end` goes 64 bytes per step. */
__device__ inline uint32_t count_long(const uint8_t *fa, const uint8_t *fb, const uint8_t *end, uint32_t lane, uint32_t done)
{
    while (fa + done + 2048u <= end) {
        uint4 a0, b0, a1, b1;
        __builtin_memcpy(&a0, fa + done + 16u * lane, 16);
        __builtin_memcpy(&b0, fb + done + 16u * lane, 16);
        __builtin_memcpy(&a1, fa + done + 1024u + 16u * lane, 16);
        __builtin_memcpy(&b1, fb + done + 1024u + 16u * lane, 16);
        const uint32_t x0 = a0.x ^ b0.x, x1 = a0.y ^ b0.y, x2 = a0.z ^ b0.z, x3 = a0.w ^ b0.w;
        const uint32_t y0 = a1.x ^ b1.x, y1 = a1.y ^ b1.y, y2 = a1.z ^ b1.z, y3 = a1.w ^ b1.w;
        const unsigned long long m0 = wave_ballot((x0 | x1 | x2 | x3) != 0u), m1 = wave_ballot((y0 | y1 | y2 | y3) != 0u);
        if (m0 | m1) {
            const bool first = m0 != 0ull;
            const uint32_t f = ctz64(first ? m0 : m1);
            const uint32_t w0 = first ? x0 : y0, w1 = first ? x1 : y1, w2 = first ? x2 : y2, w3 = first ? x3 : y3;
            const uint32_t fd = w0 ? ((uint32_t)__builtin_ctz(w0) >> 3) : (w1 ? 4u + ((uint32_t)__builtin_ctz(w1) >> 3) : (w2 ? 8u + ((uint32_t)__builtin_ctz(w2) >> 3) : 12u + ((uint32_t)__builtin_ctz(w3 | 0x80000000u) >> 3)));
            return done + (first ? 0u : 1024u) + 16u * f + lane_get(fd, f);
        }
        done += 2048u;
    }
    for (;;) {
        const bool inb = fa + done + lane < end;
        const bool eq = inb && fa[done + lane] == fb[done + lane];
        const unsigned long long neq = wave_ballot(!eq);
        if (neq != 0ull) return done + ctz64(neq);
        done += 64u;
    }
}

/* forward and backward extension of a match in one trip to memory: bytes equal from fa/fb on (limited by
 * end) and bytes equal before ba/bb (at most blim: the library's catch-up loop); 64 bytes per step each */
__device__ inline void count_both(const uint8_t *fa, const uint8_t *fb, const uint8_t *end, const uint8_t *ba,
                                  const uint8_t *bb, uint32_t blim, uint32_t lane, uint32_t &fwd, uint32_t &back)
{
    const bool fin = fa + lane < end, bin = lane < blim;
    uint32_t x0 = 0, x1 = 1, y0 = 0, y1 = 1;
    if (fin) { x0 = fa[lane]; x1 = fb[lane]; }
    if (bin) { y0 = ba[-1 - (int)lane]; y1 = bb[-1 - (int)lane]; }
    const unsigned long long fne = wave_ballot(x0 != x1), bne = wave_ballot(y0 != y1);
    fwd = fne ? ctz64(fne) : 64u;
    back = bne ? ctz64(bne) : 64u;
    if (!fne) fwd = count_long(fa, fb, end, lane, 64u);
    if (!bne) {
        uint32_t done = 64u;
        for (;;) {
            const uint32_t k = done + lane;
            const bool eq = k < blim && ba[-1 - (int)k] == bb[-1 - (int)k];
            const unsigned long long neq = wave_ballot(!eq);
            if (neq != 0ull) { back = done + ctz64(neq); break; }
            done += 64u;
        }
    }
}


} // namespace cryo

#endif
