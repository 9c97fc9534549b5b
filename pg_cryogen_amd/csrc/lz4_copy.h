/*
 * lz4_copy.h -- the copy engine of the LZ4 block decoders: moves the bytes of up to 64 decoded sequences
 * (one per lane) into the wave's output ring.  Used by k_lz4_dec_seq (lz4_dec2.hip, sequence starts from the index)
 * and k_lz4_dec_ring (lz4_dec.hip, sequence starts from the in-wave parse).
 *
 * Round 1 moved one OUTPUT byte per lane and had to decide, per byte, whether it was a literal or a match byte
 * (24 VALU instructions per 64 bytes).  Here the kinds never meet:
 *   literals and independent matches (source ends before the batch begins; far matches -- source no longer in the
 *             ring, requested from the flushed output when the batch was decoded -- are among them): one lane per
 *             sequence copies its run 16 bytes per step with unaligned wide LDS accesses, exact to the byte; no
 *             dependencies, a few LDS round trips per batch whatever its size;
 *   dependent matches: "match space" = their bytes concatenated; byte m -> match r(m) (bitmap of match starts +
 *             v_mbcnt) -> destination, source = destination - offset, one byte per lane.  Chunks of 64 match bytes
 *             are produced in order, so only a source inside the chunk's own span can be unready (frontier rounds).
 */
#ifndef CRYO_LZ4_COPY_H
#define CRYO_LZ4_COPY_H

#include "lz_common.h"

namespace cryo {
namespace {

#define LDS_FENCE() asm volatile("" ::: "memory")

template <uint32_t R, uint32_t TMAX>
struct CopyLds {
    static constexpr uint32_t kBmW = TMAX / 32; /* bitmap words of match space      */
    static constexpr uint32_t kNc = TMAX / 64;  /* chunks of match space (at most)  */
    static constexpr uint32_t kWords = kBmW + kNc;
    uint8_t *ring;                 /* R + 16: output ring; a literal piece that runs over the end lands in the 16 extra bytes and is folded back */
    uint8_t *in;                   /* kInRing + 16: input ring; the extra bytes mirror its first 16 (16-byte reads near the end) */
    uint32_t *mmeta;               /* 64: dependent match -> (start of its match inside the batch minus its position in match space: 11 bits) | offset << 11 */
    uint32_t *mbm;                 /* kBmW + kNc: bitmap of match starts in match space, then per-chunk bases */
};

/*
 * One run per lane, 16 bytes per step, exact to the byte: `rem` bytes from `src` (a pointer into LDS, or -- FAR -- the
 * registers xa, xb) to ring position `dv`.  Unaligned wide LDS accesses are exact on gfx950 and cost about one LDS
 * cycle per active lane whatever their width (profiles/microbench).  The rest of a run goes out as 8/4/2/1-byte
 * pieces: a byte too many would land in the next run.  A piece that runs over the ring's end lands in the 16
 * bytes behind it; `spill` counts them (folded back by the caller).
 */
template <uint32_t R, uint32_t SMASK>
__device__ inline void lane_runs(uint8_t *ring, const uint8_t *sbase, uint32_t rem, uint32_t sv, uint32_t dv, const bool far,
                                 const uint4 xa, const uint4 xb, uint32_t &spill)
{
    /* (the loop's variables advance by a selected step, outside the `if`: with the updates inside it the compiler carried two
     * sets of register copies through every turn, 10 of its 22 vector instructions; k_lz4_dec_seq is bound by their issue) */
    uint32_t it = 0; /* bytes moved so far */
    bool go = rem >= 16u;
    if (wave_any(go)) {
        do {
            const uint32_t di = dv & (R - 1u);
            if (go) {
                uint4 v = it == 0u ? xa : xb;
                if (!far) __builtin_memcpy(&v, sbase + (sv & SMASK), 16);
                __builtin_memcpy(ring + di, &v, 16);
            }
            if (go && di + 16u > R) spill = di + 16u - R;
            const uint32_t step = go ? 16u : 0u;
            sv += step; dv += step; rem -= step; it += step;
            go = rem >= 16u;
        } while (wave_any(go));
    }
    /* what is left of a run that already moved 16 bytes: one more 16-byte copy that ends where the run ends (it
     * rewrites a few bytes with the same data) instead of up to four exact pieces */
    const bool lap = rem != 0u && it != 0u && !far;
    if (lap) {
        uint4 v;
        __builtin_memcpy(&v, sbase + ((sv + rem - 16u) & SMASK), 16);
        const uint32_t di = (dv + rem - 16u) & (R - 1u);
        __builtin_memcpy(ring + di, &v, 16);
        if (di + 16u > R) spill = di + 16u - R;
        rem = 0u;
    }
    if (rem != 0u) {
        uint4 v = it == 0u ? xa : xb;
        if (!far) __builtin_memcpy(&v, sbase + (sv & SMASK), 16);
        uint32_t di = dv & (R - 1u);
        if (di + rem > R) spill = di + rem - R;
        if (rem & 8u) { __builtin_memcpy(ring + di, &v.x, 8); di += 8u; v.x = v.z; v.y = v.w; }
        if (rem & 4u) { __builtin_memcpy(ring + di, &v.x, 4); di += 4u; v.x = v.y; }
        if (rem & 2u) { const uint16_t h = (uint16_t)v.x; __builtin_memcpy(ring + di, &h, 2); di += 2u; v.x >>= 16; }
        if (rem & 1u) ring[di] = (uint8_t)v.x;
    }
}


/* the 8/4/2/1-byte pieces of a run shorter than 16 bytes, from registers (exact to the byte) */
template <uint32_t R>
__device__ inline void lane_tail_pieces(uint8_t *ring, uint4 v, const uint32_t rem, uint32_t di, uint32_t &spill)
{
    if (di + rem > R) spill = di + rem - R;
    if (rem & 8u) { __builtin_memcpy(ring + di, &v.x, 8); di += 8u; v.x = v.z; v.y = v.w; }
    if (rem & 4u) { __builtin_memcpy(ring + di, &v.x, 4); di += 4u; v.x = v.y; }
    if (rem & 2u) { const uint16_t h = (uint16_t)v.x; __builtin_memcpy(ring + di, &h, 2); di += 2u; v.x >>= 16; }
    if (rem & 1u) ring[di] = (uint8_t)v.x;
}
template <uint32_t R, uint32_t N, typename T>
__device__ inline void lane_store(uint8_t *ring, const uint32_t dv, const T &v, uint32_t &spill)
{
    const uint32_t di = dv & (R - 1u);
    __builtin_memcpy(ring + di, &v, N);
    if (di + N > R) spill = di + N - R;
}

/*
 * The lane-per-sequence copies of a batch, round 4 form.  A wave's time per batch is a chain of dependent LDS round
 * trips (~300 cycles each with 24 waves per CU on the LDS), not instructions: the loop below it (lane_runs: read 16,
 * wait, write 16, again; then the same for the matches) took three to five trips for runs of 20-32 bytes and matches of
 * 5-12.  Here every read of the common shapes -- a literal run's first 16 bytes, its last 16 when it has 17..32, an
 * independent match's first 16 -- is issued before the first write: one trip.  Writes are exact to the byte: a run of
 * 17..32 is two 16-byte stores that overlap, a match of 8..15 two 8-byte stores, of 4..7 two 4-byte stores (LZ4 has no
 * shorter match; zstd's 3-byte ones take the piece ladder).  Whatever is longer goes on from byte 16 in lane_runs.
 */
template <uint32_t R>
__device__ inline void lane_copies_v2(uint8_t *ring, const uint8_t *in, const uint32_t L, const uint32_t lpos, const uint32_t dl,
                                      const uint32_t ML, const uint32_t msrc, const uint32_t dm, const bool isfar,
                                      const uint4 xfa, const uint4 xfb, uint32_t &spill)
{
    uint4 la = make_uint4(0, 0, 0, 0), lc = la, ma = xfa;
    const bool l2 = L > 16u && L <= 32u;
    if (L != 0u) __builtin_memcpy(&la, in + (lpos & kInMask), 16);
    if (l2) __builtin_memcpy(&lc, in + ((lpos + L - 16u) & kInMask), 16);
    if (ML != 0u && !isfar) __builtin_memcpy(&ma, ring + (msrc & (R - 1u)), 16);
    /* literals */
    if (L >= 16u) lane_store<R, 16>(ring, dl, la, spill);
    if (l2) lane_store<R, 16>(ring, dl + L - 16u, lc, spill);
    /* matches */
    if (ML == 16u || (ML > 16u && isfar)) lane_store<R, 16>(ring, dm, ma, spill);
    if (ML >= 8u && ML < 16u) {
        const uint32_t s = ML - 8u; /* bytes s .. s+7 of the sixteen */
        const bool hi = s >= 4u;
        const uint32_t a = hi ? ma.y : ma.x, b = hi ? ma.z : ma.y, c = hi ? ma.w : ma.z;
        const uint2 head = make_uint2(ma.x, ma.y);
        const uint2 tail = make_uint2(__builtin_amdgcn_alignbyte(b, a, s), __builtin_amdgcn_alignbyte(c, b, s));
        lane_store<R, 8>(ring, dm, head, spill);
        lane_store<R, 8>(ring, dm + s, tail, spill);
    }
    if (ML >= 4u && ML < 8u) {
        const uint32_t s = ML - 4u;
        const uint32_t tail = __builtin_amdgcn_alignbyte(ma.y, ma.x, s);
        lane_store<R, 4>(ring, dm, ma.x, spill);
        lane_store<R, 4>(ring, dm + s, tail, spill);
    }
    /* the rare shapes */
    const bool lshort = L != 0u && L < 16u, mshort = ML != 0u && ML < 4u;
    if (wave_any(lshort | mshort)) {
        if (lshort) lane_tail_pieces<R>(ring, la, L, dl & (R - 1u), spill);
        if (mshort) lane_tail_pieces<R>(ring, ma, ML, dm & (R - 1u), spill);
    }
    if (wave_any(L > 32u)) {
        const uint4 z = make_uint4(0, 0, 0, 0);
        lane_runs<R, kInMask>(ring, in, L > 32u ? L - 16u : 0u, lpos + 16u, dl + 16u, false, z, z, spill);
    }
    if (wave_any(ML > 16u)) { /* a far match is at most 32 bytes: its second half is xfb; a near one starts over (16-byte steps) */
        const uint32_t done = isfar ? 16u : 0u;
        lane_runs<R, R - 1u>(ring, ring, ML > 16u ? ML - done : 0u, msrc + done, dm + done, isfar, xfb, xfb, spill);
    }
}

/*
 * Copy the bytes of up to 64 sequences (lane i < nseq holds sequence i).
 *   ostart: first output byte of the sequence inside the batch; ll literal bytes from virtual input position
 *   lpos, then ml match bytes at distance off (a match with off < ml overlaps itself: it is always a dependent one
 *   and the frontier rounds take it off bytes at a time; callers keep those short); T = total bytes <= TMAX.
 *   isfar: the match's source is no longer in the ring; its bytes are in xfa/xfb (requested by the caller).
 */
/*   NOLIT: the literal runs are somebody else's (the dual-wave decoder, lz4_dec2.hip: the block's other wave copies them
 *   while this one still works on the batch before); ll only places the matches. */
template <uint32_t R, uint32_t TMAX, bool NOLIT = false>
__device__ inline void seq_copy(Wave<R> &w, const CopyLds<R, TMAX> &L, const uint32_t nseq, const uint32_t ostart,
                                const uint32_t ll, const uint32_t ml, const uint32_t off, const uint32_t lpos,
                                const uint32_t T, const bool isfar, const uint4 xfa, const uint4 xfb, Stats &st)
{
    static_assert(R - TMAX >= TMAX + 1023u, "an offset must be either in the ring or flushed");
    constexpr uint32_t kBmW = CopyLds<R, TMAX>::kBmW, kNc = CopyLds<R, TMAX>::kNc;
    const uint32_t lane = w.lane;
    const uint32_t op0 = w.op;
    const bool act = lane < nseq;
    st.batches++;
    st.batch_seqs += nseq;

    /* A match whose source ends before the batch begins depends on nothing the batch produces (about half of them
     * on tuple data, far matches included): those go lane-per-sequence like the literals.  The others form the
     * "match space". */
    const uint32_t mrel = ostart + ll;                           /* match start inside the batch */
    const bool indep = act && (isfar || off >= mrel + ml);
    const bool dep = act && !indep;
    const uint32_t mlx = dep ? ml : 0u;
    const uint32_t mend = scan64_incl(mlx);
    const uint32_t mcum = mend - mlx;                            /* dependent match bytes before this sequence's */
    const uint32_t MT = lane_get(mend, 63u);
    const unsigned long long depm = wave_ballot(dep);
    const uint32_t drank = __builtin_amdgcn_mbcnt_hi((uint32_t)(depm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)depm, 0u));

    {
    if (lane < kBmW) L.mbm[lane] = 0u;
    LDS_FENCE();
    if (dep) {
        /* bit (start - 1) for every dependent match that starts at a position >= 1 of match space: the number of set
         * bits BELOW a position is the rank of the match it belongs to.
         * meta: x = output position of the match minus its position in match space; y = offset */
        static_assert(TMAX < 2048u, "11 bits");
        L.mmeta[drank] = (mrel - mcum) | (off << 11); /* off < 2^21: LZ4's 16 bits, zstd's window */
        if (mcum != 0u) atomicOr(&L.mbm[(mcum - 1u) >> 5], 1u << ((mcum - 1u) & 31u));
    }
    LDS_FENCE();
    }
    stamp(st, 4);

    /* ---- literals and independent matches: one lane per sequence ---- */
    {
        uint32_t spill = 0; /* bytes this lane wrote beyond the ring's end */
        const uint4 z = make_uint4(0, 0, 0, 0);
        if (!NOLIT) lane_runs<R, kInMask>(L.ring, L.in, act ? ll : 0u, lpos, op0 + ostart, false, z, z, spill);
        /* (the 16 bytes behind the ring mirror its first 16 for reads that start in its last 15: lz4_seq_batch) */
        lane_runs<R, R - 1u>(L.ring, L.ring, indep ? ml : 0u, op0 + mrel - off, op0 + mrel, isfar, xfa, xfb, spill);
        /* a batch crosses the ring's end at most once: fold the bytes that ran over back to the start */
        const unsigned long long sm = wave_ballot(spill != 0u);
        if (sm != 0ull) {
            LDS_FENCE();
            const uint32_t k = lane_get(spill, ctz64(sm));
            if (lane < k) L.ring[lane] = L.ring[R + lane];
        }
    }
    LDS_FENCE();
    stamp(st, 5);
    {
        /* bits before each match-space chunk */
        static_assert(kNc <= 64, "one lane per chunk");
        uint32_t cnt = 0;
        if (lane < kNc) {
            const uint2 wv = *reinterpret_cast<const uint2 *>(&L.mbm[lane * 2u]);
            cnt = (uint32_t)(__popc(wv.x) + __popc(wv.y));
        }
        const uint32_t exc = scan64_incl(cnt) - cnt;
        if (lane < kNc) L.mbm[kBmW + lane] = exc;
    }
    LDS_FENCE();

    /* ---- match space: chunks in order ---- */
    {
        const uint32_t nM = (MT + 63u) >> 6;
        constexpr uint32_t U = 4;
        for (uint32_t c0 = 0; c0 < nM; c0 += U) {
            uint32_t da[U], ra[U];
            bool pendv[U], actv[U];
            unsigned long long pmv[U]; /* pendv as a lane mask, built from the votes of the plain compares: a vote on a
                                        * combined predicate costs two vector instructions more (v_cndmask + v_cmp) */
            {
                uint32_t idx[U];
                uint32_t mt[U];
#pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    const uint32_t c = c0 + u < kNc ? c0 + u : kNc - 1u;
                    const uint2 wv = *reinterpret_cast<const uint2 *>(&L.mbm[c * 2u]);
                    const uint32_t bc = L.mbm[kBmW + c];
                    idx[u] = (bc + __builtin_amdgcn_mbcnt_hi(wv.y, __builtin_amdgcn_mbcnt_lo(wv.x, 0u))) & 63u;
                }
#pragma unroll
                for (uint32_t u = 0; u < U; u++) mt[u] = L.mmeta[idx[u]];
#pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    const uint32_t m = (c0 + u) * 64u + lane;
                    const bool a = m < MT;
                    da[u] = op0 + m + (mt[u] & 2047u);    /* absolute output position of this byte */
                    ra[u] = da[u] - (mt[u] >> 11);            /* ... and of its source                  */
                    const uint32_t d0 = uni(da[u]);       /* first byte of the chunk               */
                    const bool ge = ra[u] >= d0;
                    actv[u] = a;
                    pendv[u] = a && ge;
                    pmv[u] = wave_ballot(a) & wave_ballot(ge);
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                if ((c0 + u) * 64u < MT) {
                    st.chunks++;
                    uint8_t *dp = &L.ring[da[u] & (R - 1u)];
                    const uint8_t *sp = &L.ring[ra[u] & (R - 1u)];
                    const uint32_t x = *sp;
                    if (actv[u]) *dp = (uint8_t)x;
                    /* a source inside this chunk's own span may not be written yet (or be a literal, which is):
                     * everything below the first pending byte is final, so a pending byte whose source lies below
                     * it can be taken; the first pending byte itself always can */
                    bool pend = pendv[u];
                    unsigned long long pm = pmv[u];
                    while (pm != 0ull) {
                        st.rounds++;
                        const uint32_t F = lane_get(da[u], ctz64(pm));
                        const bool lt = ra[u] < F; /* the first pending lane too: its source lies below itself */
                        const unsigned long long ltm = wave_ballot(lt); /* right behind the compare: it is the compare's mask then */
                        if (pend && lt) *dp = *sp;
                        pm &= ~ltm;
                        pend = pend && !lt;
                    }
                }
            }
        }
    }
    stamp(st, 6);
    w.op = op0 + T; /* flushed by the next batch (or the general path), together with its own requests */
}


} // namespace
} // namespace cryo

#endif
