/*
 * lz4_dec2.hip -- LZ4 block decode from a sequence index: one sequence per lane + run-space copy engine.
 *
 * Replaces LZ4_decompress_safe(compressed, out, compressed_size, CRYO_BLCKSZ) (reference
 * compression.c:84) for a batch of independent blocks, with the accept/reject rules listed at the top of
 * lz4_dec.hip.  Two kernels:
 *
 *   k_lz4_index   (lz4_index.hip) one LANE per walker, one or several walkers per block: walks the token chain and
 *                 writes the low 16 bits of every token position to the block's row of the workspace (the serial
 *                 part of LZ4 decoding, run for all blocks of the batch at once);
 *   k_lz4_dec_seq one WAVE per block: lane i loads position i of the row, decodes "its" sequence
 *                 (literal length, offset, match length), a wave scan gives output positions, and the
 *                 bytes are moved by the run-space copy engine below.  The row is only a hint: every
 *                 batch checks it against the stream (lz4_seq_batch), so a wrong row costs speed, never bytes.
 *
 * Copy engine (seq_copy).  Round 1 moved one OUTPUT byte per lane and had to decide, per byte, whether it was a
 * literal or a match byte (pass A: 24 VALU instructions per 64 bytes).  Here the kinds never meet:
 *   literals and independent matches (source ends before the batch begins; far matches -- source no longer in
 *             the ring, requested from the flushed output when the batch was decoded -- are among them): one lane
 *             per sequence copies its run 16 bytes per step with unaligned wide LDS accesses, exact to the byte;
 *             no dependencies, a few LDS round trips per batch whatever its size;
 *   dependent matches: "match space" = their bytes concatenated; byte m -> match r(m) (bitmap of match starts +
 *             v_mbcnt) -> destination, source = destination - offset, one byte per lane.  Chunks of 64 match bytes
 *             are produced in order, so only a source inside the chunk's own span can be unready (frontier rounds).
 */
#include "lz_common.h"
#include "lz4_copy.h"
#include <cstdio>
#include <cstdlib>

namespace cryo {

/* ---------------------------------------------------------------------------------------------
 * Decoder
 * --------------------------------------------------------------------------------------------- */

constexpr uint32_t kSeqWin = 1536;  /* compressed bytes a batch may span (in_hi stays < vp + kInRing)   */
constexpr uint32_t kT2 = 1536;      /* output bytes per batch: the largest T with R - T >= T + 1023 at R = 4096, i.e. every
                                     * offset is either still in the ring (off < R - T) or flushed (off >= T + 1023) */

/*
 * The block's row of the index, as the walkers left it (lz4_index.hip): per segment an "extension" piece (what the left
 * neighbour's walker visited before the chains met) and the segment's own records from the meeting point on.  Lane s
 * holds segment s's descriptor; the pieces are consumed in order.  A batch takes its 64 entries from the current piece and,
 * where that ends, from the two behind it -- the tail of a segment's records, the (short) extension into the next segment
 * and that segment's own records (round 4: a batch used to stop at every piece's end -- with 32 walkers per block that is 64
 * short batches on top of a 128 KiB block's 100, and the reason mid-sized batches could not use more walkers).
 */
template <bool MULTI> /* false: one walker per block, the row is one piece (the headline batch: nothing of the window below) */
struct IndexRow {
    const uint16_t *base;      /* the block's row */
    uint2 sd;                  /* lane s: descriptor of segment s */
    uint32_t cap_s, ext, npieces, pc; /* pc: the next piece to look at */
    const uint16_t *t0, *t1, *t2; /* the current piece and the next two that are not empty */
    uint32_t len0, len1, len2, n0; /* their entries; entries of the current piece used so far */

    __device__ inline void find(const uint16_t *&t, uint32_t &len)
    {
        len = 0;
        t = base;
        while (pc < npieces) {
            const uint32_t sg = pc >> 1;
            const uint32_t dx = lane_get(sd.x, sg), dy = lane_get(sd.y, sg);
            const uint32_t l = (pc & 1u) ? dy : (dx & 0xffffu);
            const uint32_t b = sg * cap_s + ((pc & 1u) ? ext + (dx >> 16) : 0u);
            pc++;
            if (l != 0u) { t = base + b; len = l; break; }
        }
    }
    __device__ inline void open(const uint16_t *row, const uint2 *seg, const uint64_t blk, const uint32_t logS, const uint32_t cap_s_,
                                const uint32_t ext_, const uint32_t lane)
    {
        base = row;
        cap_s = cap_s_; ext = ext_;
        npieces = 2u << logS;
        sd = make_uint2(0, 0);
        if (lane < (1u << logS)) sd = seg[(blk << logS) + lane];
        pc = 0; n0 = 0;
        find(t0, len0);
        t1 = t2 = base; len1 = len2 = 0;
        if (MULTI) { find(t1, len1); find(t2, len2); }
    }
    /* entries from position k behind the cursor on: how many there are (64 at most), and this lane's */
    __device__ inline uint32_t avail(const uint32_t k) const
    {
        const uint32_t have = MULTI ? len0 + len1 + len2 : len0, at = n0 + k;
        return at < have ? (have - at < 64u ? have - at : 64u) : 0u;
    }
    __device__ inline uint32_t entry(const uint32_t k, const uint32_t lane) const
    {
        const uint32_t i = n0 + k + lane;
        uint32_t e = 0;
        if (MULTI) {
            const uint16_t *q = i < len0 ? t0 + i : (i < len0 + len1 ? t1 + (i - len0) : t2 + (i - len0 - len1));
            if (i < len0 + len1 + len2) e = *q;
        } else if (i < len0) e = t0[i];
        return e;
    }
    __device__ inline void advance(const uint32_t n)
    {
        n0 += n;
        while (MULTI && len0 != 0u && n0 >= len0) { /* the current piece is used up: the next one becomes it */
            n0 -= len0;
            t0 = t1; len0 = len1;
            t1 = t2; len1 = len2;
            find(t2, len2);
        }
    }
};

/* lane i gets lane i+1's value (lane 63: 0) */
__device__ inline uint32_t lane_next(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

/*
 * One batch: sequences n0 .. n0+63 of the block start at the positions in the index row (epos = this lane's
 * entry).  Returns the number of sequences decoded (0: the caller takes one sequence through the general path).
 */
template <uint32_t R, bool MULTI>
__device__ inline uint32_t lz4_seq_batch(Wave<R> &w, const CopyLds<R, kT2> &L, uint32_t &vp, const uint32_t B,
                                         const uint32_t epos, const uint32_t navail, const IndexRow<MULTI> &ix, uint32_t &epre, Stats &st,
                                         unsigned long long *stop_hist = nullptr)
{
    const uint32_t lane = w.lane;
    const uint32_t vend = w.vend;
    if (vend < 32u || B < 32u || navail == 0u) return 0;
    const uint32_t vsafe = vend - 16u;
    if (vp + 64u > vsafe || w.op + 64u > B) return 0;
    stamp(st, 7);
    /* Vector-memory operations of a batch, in issue order: [start] none -- the staging below only writes chunks that
     * were requested in the middle of the previous batch -- then, once the batch is decoded: the output stores of
     * the previous batch, the next input chunks, the next batch's index entries, this batch's far sources.  The one
     * wait that follows them (the far sources, seq_copy) comes after the literal copy.  vmcnt counts in order, so a
     * request issued just before a wait is a full trip to memory on the wave's critical path: with the requests
     * at the end of a batch and the waits at its start, 32 % of the wave's cycles were that. */
    while (w.in_hi < vend && w.in_hi < vp + kSeqWin) w.refill_deferred();
    LDS_FENCE();
    /* 16-byte literal reads may start in the ring's last 15 bytes: keep a copy of its first 16 behind it */
    uint2 mir = make_uint2(0, 0);
    if (lane < 2u) mir = *reinterpret_cast<const uint2 *>(L.in + lane * 8u);
    else if (lane < 4u) mir = *reinterpret_cast<const uint2 *>(L.ring + (lane - 2u) * 8u); /* the output ring likewise */
    stamp(st, 0);
    /* positions: the row holds the low 16 bits of the offset in the compressed block; a batch spans < 64 KiB */
    const uint32_t pos = vp + ((epos + w.delta - vp) & 0xffffu);
    const bool cand = lane < navail && pos + 8u <= vp + kSeqWin; /* also keeps the reads below inside the staged window */
    const uint32_t rp = cand ? pos : vp;
    const uint32_t t = L.in[rp & kInMask];
    const uint32_t e1 = L.in[(rp + 1u) & kInMask];
    uint32_t ll = t >> 4;
    uint32_t k = 1u;
    if (ll == 15u) { ll += e1; k = 2u; }
    const uint32_t q = rp + k + ll; /* offset field */
    const bool inwin = q + 8u <= vp + kSeqWin;
    const uint32_t rq = inwin ? q : vp;
    const uint32_t off = (uint32_t)L.in[rq & kInMask] | ((uint32_t)L.in[(rq + 1u) & kInMask] << 8);
    const uint32_t e2 = L.in[(rq + 2u) & kInMask];
    if (lane < 2u) *reinterpret_cast<uint2 *>(L.in + kInRing + lane * 8u) = mir;
    else if (lane < 4u) *reinterpret_cast<uint2 *>(L.ring + R + (lane - 2u) * 8u) = mir;
    uint32_t ml = (t & 15u) + 4u;
    uint32_t dlen = k + ll + 2u;
    const bool hasM = (t & 15u) == 15u;
    if (hasM) { ml += e2; dlen += 1u; }
    const uint32_t outlen = cand ? ll + ml : 0u;
    const uint32_t oend = scan64_incl(outlen);
    const uint32_t ostart = oend - outlen;
    const uint32_t mabs = w.op + ostart + ll;
    /* the row is a hint: lane 0 must sit on the wave's own position and every sequence must end where the
     * next lane's starts (the last candidate needs no successor) */
    const uint32_t npos = lane_next(pos);
    const bool chain = (lane == 0u ? pos == vp : true) && (lane + 1u >= navail || lane == 63u || npos == pos + dlen);
    /* far matches (source older than the ring can still hold when the batch is done; already flushed): two 16-byte
     * requests per lane */
    constexpr uint32_t kNear = R - kT2;
    /* a short match that overlaps itself (runs: offset 1..3) stays in the batch: it is a dependent match, and the
     * frontier rounds of match space resolve it off bytes per round.  Leaving it to the general path cost a cut
     * batch, an empty one and the wave-uniform parser, 12 times per block on tuple data. */
    constexpr uint32_t kOvlMax = 64;
    const bool isfar = cand && off >= kNear;
    const bool ok = cand && inwin && chain && !(t >= 0xf0u && e1 == 255u) && !(hasM && e2 == 255u) && (off >= ml || (off != 0u && ml <= kOvlMax)) && off <= mabs &&
                    pos + dlen <= vsafe && oend <= kT2 && w.op + oend + 16u <= B && !(isfar && ml > 32u);
    const unsigned long long badmask = wave_ballot(!ok);
    const uint32_t nseq = badmask ? ctz64(badmask) : 64u;
    if (st.on && stop_hist && badmask && nseq < navail && lane == nseq) { /* why the batch stops in front of this sequence */
        const int why = !cand ? 0 : !inwin ? 1 : !chain ? 2 : (t >= 0xf0u && e1 == 255u) ? 3 : (hasM && e2 == 255u) ? 4 : (off < ml && (off == 0u || ml > kOvlMax)) ? 5 : off > mabs ? 6
                        : pos + dlen > vsafe ? 7 : oend > kT2 ? 8 : w.op + oend + 16u > B ? 9 : 10;
        atomicAdd(&stop_hist[why], 1ull);
    }
    stamp(st, 1);
    if (!(st.ablate & 2u)) w.flush();   /* what earlier batches produced; far sources below are read back from it */
    else w.flushed = w.op & ~(kChunk - 1u);
    stamp(st, 2);
    if (!(st.ablate & 4u)) w.top_up(); else w.nstale = 0;
    if (nseq == 0u) return 0;
    const uint32_t T = lane_get(oend, nseq - 1u);
    const uint32_t used = lane_get(pos + dlen, nseq - 1u) - vp;
    /* the next batch's positions are requested now: their trip to memory hides behind this batch's copy */
    epre = ix.entry(nseq, lane);
    /* ... and so are the sources of its far matches (flushed output: off >= T + 1023 behind a match).  (Round 4 tried
     * them first, as assembly loads with a counted wait behind the literal copy instead of the compiler's vmcnt(0):
     * 3 % slower -- the kernel is bound by instruction issue and LDS cycles, not by this wait;
     * profiles/r04_lz4_decode_ab.txt.) */
    uint4 xfa, xfb; /* only a far lane's are looked at: whatever the registers hold will do for the others (eight moves less) */
    asm volatile("" : "=v"(xfa.x), "=v"(xfa.y), "=v"(xfa.z), "=v"(xfa.w), "=v"(xfb.x), "=v"(xfb.y), "=v"(xfb.z), "=v"(xfb.w));
    if (lane < nseq && isfar && !(st.ablate & 1u)) {
        const uint8_t *g = w.dst + (mabs - off);
        __builtin_memcpy(&xfa, g, 16);
        __builtin_memcpy(&xfb, g + 16, 16);
    }
    stamp(st, 3);
    seq_copy<R, kT2>(w, L, nseq, ostart, ll, ml, off, pos + k, T, isfar, xfa, xfb, st);
    vp += used;
    stamp(st, 0);
    return nseq;
}

/*
 * One sequence through the wave-uniform parser: every token form (255-runs, literal runs and matches of any
 * length, overlapping matches, the block's last sequence) and every reject rule.  Returns 0 = decoded,
 * 1 = decoded and it was the block's last sequence, 2 = malformed.
 */
template <uint32_t R>
__device__ inline uint32_t lz4_general_seq(Wave<R> &w, uint32_t &vp, const uint32_t B)
{
    const uint32_t lane = w.lane, vend = w.vend;
    w.flush();
    w.need(vp);
    uint32_t win = w.window(vp);

    /* ---- token and literal length (window lane 0 = token) ---- */
    const uint32_t token = lane_get(win, 0);
    uint32_t ll = token >> 4;
    uint32_t k = 1; /* window lane of the first byte after the literal-length field */
    if (ll == 15u) {
        /* extension bytes: 255 ... 255 x ; find the terminating byte with a ballot */
        uint32_t wbase = vp;
        for (;;) {
            const unsigned long long m = wave_ballot(win != 255u) & ~((1ull << k) - 1ull);
            if (m != 0ull) {
                const uint32_t f = ctz64(m);
                ll += (f - k) * 255u + lane_get(win, f);
                k = f + 1u;
                break;
            }
            /* a 255 at position q is only legal while q + 16 < csize */
            if (wbase + 63u + 16u >= vend) return 2u;
            ll += (64u - k) * 255u;
            wbase += 64u;
            w.need(wbase);
            win = w.window(wbase);
            k = 0;
        }
        /* terminating byte at position pe must satisfy pe + 15 < csize */
        if (wbase + (k - 1u) + 15u >= vend) return 2u;
        vp = wbase; /* lane k of the current window is the byte after the field */
    }
    const uint32_t lp = vp + k; /* virtual position of the first literal */

    /* ---- literal run ---- */
    const bool last = (w.op + ll + 12u > B) || (lp + ll + 8u > vend);
    if (last && (lp + ll != vend || w.op + ll > B)) return 2u;
    {
        /* first piece straight from the window registers */
        uint32_t n0 = 64u - k;
        if (n0 > ll) n0 = ll;
        if (lane >= k && lane < k + n0) w.ring[(w.op + lane - k) & (R - 1)] = (uint8_t)win;
        w.op += n0;
        uint32_t rem = ll - n0;
        uint32_t p = lp + n0;
        while (rem) {
            w.flush();
            if (wave_stream_literals(w, p, rem)) continue; /* long run at a 1 KiB boundary: lz_common.h */
            w.need(p);
            const uint32_t x = w.window(p);
            uint32_t n = rem < 64u ? rem : 64u;
            if (rem >= 3u * R) { const uint32_t to = kChunk - (w.op & (kChunk - 1u)); n = n < to ? n : to; } /* land on the boundary */
            if (lane < n) w.ring[(w.op + lane) & (R - 1)] = (uint8_t)x;
            w.op += n;
            p += n;
            rem -= n;
        }
        if (last) return 1u;
        /* keep the 2 offset bytes and the first length byte inside the window */
        if (p - vp > 61u) {
            vp = p;
            w.need(vp);
            win = w.window(vp);
        }
        k = p - vp; /* window lane of the offset's low byte */
    }

    /* ---- offset and match length ---- */
    const uint32_t off = lane_get(win, k) | (lane_get(win, k + 1u) << 8);
    uint32_t ml = token & 15u;
    k += 2u;
    if (ml == 15u) {
        uint32_t wbase = vp;
        for (;;) {
            const unsigned long long m = (k < 64u) ? (wave_ballot(win != 255u) & ~((1ull << k) - 1ull)) : 0ull;
            if (m != 0ull) {
                const uint32_t f = ctz64(m);
                ml += (f - k) * 255u + lane_get(win, f);
                k = f + 1u;
                break;
            }
            /* every extension byte at position q needs q + 5 < csize */
            if (wbase + 63u + 5u >= vend) return 2u;
            ml += (64u - k) * 255u;
            wbase += 64u;
            w.need(wbase);
            win = w.window(wbase);
            k = 0;
        }
        if (wbase + (k - 1u) + 5u >= vend) return 2u;
        vp = wbase;
    }
    ml += 4u;
    if (off > w.op || w.op + ml + 5u > B) return 2u;
    vp += k;
    wave_copy_match(w, off, ml); /* lz_common.h: near / far / overlapping / offset 0 */
    return 0u;
}

constexpr uint32_t kDecR = 4096;   /* output ring of a decoding wave */
constexpr uint32_t kDecWpb = 4;   /* waves (blocks) per workgroup */
constexpr int kDecOcc = 6;        /* workgroups per CU the register allocator aims at */

template <uint32_t R, bool STATS, uint32_t WPB, bool MULTI>
__global__ void __launch_bounds__(64 * WPB, kDecOcc)
k_lz4_dec_seq(const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
              const uint32_t *__restrict__ src_size, uint8_t *dst_base, uint64_t dst_stride, uint32_t B,
              uint64_t n_blocks, int32_t *__restrict__ status, unsigned long long *stats,
              const uint16_t *__restrict__ tbl, uint32_t tbl_cap, const uint2 *__restrict__ seg, const uint32_t logS,
              const uint32_t cap_s, const uint32_t ext, const uint32_t skip_heavy, const uint32_t *__restrict__ decoded, const uint64_t blk0)
{
    Stats st = {};
    st.on = STATS;
    if (STATS) { st.ablate = (uint32_t)stats[7]; st.t0 = __builtin_amdgcn_s_memtime(); }
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[WPB][R + 16];
    __shared__ __attribute__((aligned(16))) uint8_t s_in[WPB][kInRing + 16];
    __shared__ __attribute__((aligned(8))) uint32_t s_mmeta[WPB][64]; /* packed: 26 880 bytes per workgroup of four waves at R = 4096, six per CU */
    __shared__ __attribute__((aligned(8))) uint32_t s_mbm[WPB][CopyLds<R, kT2>::kWords];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = uni(threadIdx.x >> 6);
    const uint64_t blk = blk0 + (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid; /* blocks blk0 .. n_blocks - 1 are this launch's */
    if (blk >= n_blocks) return;

    const uint8_t *base = src_base + uni64(src_off[blk]);
    const uint32_t csize = uni(src_size[blk]);
    if (skip_heavy != 0u && lz4_literal_heavy(csize, B)) return; /* left to the in-wave parser (kernels.h) */
    if (decoded != nullptr && uni(decoded[blk]) != 0u) return;        /* by the few-blocks path (lz4_lat.hip) */

    Wave<R> w;
    const CopyLds<R, kT2> L = {s_ring[wid], s_in[wid], s_mmeta[wid], s_mbm[wid]};
    w.ring = L.ring;
    w.in = L.in;
    w.lane = lane;
    w.delta = (uint32_t)(reinterpret_cast<uintptr_t>(base) & 127u); /* staging chunks are whole lines */
    w.abase = base - w.delta;
    w.vend = w.delta + csize;
    w.in_hi = 0;
    w.dst = dst_base + uni64(blk * dst_stride);
    w.dst_aligned = (reinterpret_cast<uintptr_t>(w.dst) & 15u) == 0;
    w.op = 0;
    w.flushed = 0;

    uint32_t vp = w.delta;
    bool bad = (csize == 0);
    bool done = bad;
    uint32_t skip = 0;
    IndexRow<MULTI> ix;
    ix.open(tbl + uni64(blk * (uint64_t)tbl_cap), seg, blk, logS, cap_s, ext, lane);
    uint32_t poor = 0;

    /* the first batch's index entries travel with the first input chunks (one trip to memory at the start of a block, not two) */
    uint32_t efirst = 0;
    if (!bad) efirst = ix.entry(0, lane);
    bool have_first = !bad;
    if (!bad) {
        w.prefetch();
        w.refill();
        if (w.in_hi < w.vend) w.refill();
    }
    uint32_t npre = ix.avail(0); /* entries the request on its way covers */

    while (!done) {
        if (skip == 0u) {
            uint32_t n;
            uint32_t epre = efirst;
            bool have_pre = have_first;
            bool more;
            have_first = false;
            do {
                uint32_t e = epre, navail = npre;
                if (!have_pre) { navail = ix.avail(0); e = ix.entry(0, lane); }
                n = lz4_seq_batch<R, MULTI>(w, L, vp, B, e, navail, ix, epre, st, STATS ? stats + 16 : nullptr);
                have_pre = n != 0u;
                npre = ix.avail(n);
                ix.advance(n);
                if (n == 0u) st.zero_batches++;
                more = n >= 8u || (n != 0u && n == navail); /* a batch that took all the window held may be short: go on */
            } while (more);
            /* a batch stops in front of a sequence it cannot take (overlapping match, long run, end of block):
             * that one goes through the general path and the batches resume.  Only data that keeps yielding
             * short batches (runs of overlapping matches) stays on the general path for a while. */
            poor = n < 4u ? poor + 1u : 0u;
            if (poor >= 3u) { skip = 8u; poor = 0u; }
        } else {
            skip--;
        }
        st.general_seqs++;
        ix.advance(1u);
        const uint32_t r = lz4_general_seq<R>(w, vp, B);
        if (r == 2u) { bad = true; break; }
        if (r == 1u) done = true;
    }

    if (!bad && w.op != B) bad = true;
    if (!bad) {
        w.flush();
        w.flush_tail();
    }
    if (lane == 0) status[blk] = bad ? CRYO_ST_CORRUPT : CRYO_ST_OK;
    if (STATS && lane == 0) {
        atomicAdd(&stats[0], (unsigned long long)st.batches);
        atomicAdd(&stats[1], (unsigned long long)st.batch_seqs);
        atomicAdd(&stats[2], (unsigned long long)st.general_seqs);
        atomicAdd(&stats[3], (unsigned long long)st.chunks);
        atomicAdd(&stats[4], (unsigned long long)st.rounds);
        atomicAdd(&stats[5], (unsigned long long)st.zero_batches);
        stamp(st, 7);
        for (int k = 0; k < 8; k++) atomicAdd(&stats[8 + k], st.t[k]);
    }
}

/* ---------------------------------------------------------------------------------------------
 * The same decoder with TWO waves per block (round 4), for batches that do not fill the chip.
 *
 * A block is one wave's chain: 4.7 us per batch of 64 sequences with the SIMD to itself (profiles/r04_lz4_decode_ab.txt),
 * whatever else runs -- so 1 024 blocks decode no faster than 6 144, and 512 x 1 MiB take 3 ms on an idle chip.
 * The chain has two halves that touch different memory: staging, token decode, validation, requests, the LITERAL runs and
 * the FAR matches read the input ring and flushed output and write bytes no match of the batch before can touch; the near
 * independent matches and match space read and write the output ring only.  So wave A does the first half of batch n+1
 * while wave B does the second half of batch n:
 *
 *   A:  stage, decode, validate, request, far sources, literal runs, far matches of n+1;  publish the batch (LDS);
 *       barrier;  store what B finished (whole 1 KiB chunks below batch n+1)
 *   B:  barrier;  near independent matches and match space of n+1 (seq_copy<NOLIT>); never touches global memory
 *
 * One barrier per batch: A arrives with batch n+1 published, B with batch n done.  What makes the overlap safe:
 *   - the ring is 8 KiB and a match is "far" (read back from the flushed output) from R - 2T on: a ring slot B reads for
 *     batch n is never one A's batch n+1 overwrites;
 *   - A stores only what lies below the batch B is working on (complete: B arrived at the barrier), and a far source is older
 *     than that by construction (it ends 3 584 bytes or more below the batch, the stores reach to within 2 559) -- stores and
 *     far loads are the same wave's, in program order;
 *   - the 16 bytes behind the ring are the landing strip of a run that crosses the ring's end (A's literals and far matches,
 *     B's matches: two adjacent batches span less than R, so only one of them crosses) and the mirror of the ring's first
 *     bytes for B's 16-byte source reads: B refreshes it only when a source run crosses the end, which then lies below
 *     both batches;
 *   - a sequence the batches cannot take (the general path) runs on A with B parked at the barrier.
 * Verdicts and bytes are the single-wave decoder's: the same validation, the same copy engine.
 *
 * Measured (profiles/r04_lz4_dual.txt): 1 024 x 128 KiB 0.475 -> 0.34 ms, 512 x 1 MiB 2.97 -> 1.83 ms.  Each wave waits at the
 * barrier for 18-20 % of the time: a batch is as long as its longer half and the halves vary (A: 600 k busy ticks, B: 587 k,
 * lock-step total 731 k).  Moving match space's preparation from B to A (then 610 k / 606 k busy) changed nothing -- the
 * variance, not the balance, is what is left.  A queue two batches deep between the waves (16 KiB ring, counts in LDS polled
 * with s_sleep) was built and is slower everywhere: profiles/r04_lz4_dualq.txt, the patch is profiles/scripts/r04_dualq.patch.
 * --------------------------------------------------------------------------------------------- */
#ifndef CRYO_DUAL_PROF
#define CRYO_DUAL_PROF 0 /* variant builds only: block 0 prints how long each of its waves waited at the barriers */
#endif
#if CRYO_DUAL_PROF
#define DUAL_BARRIER() do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); __syncthreads(); t_wait += __builtin_amdgcn_s_memtime() - t1_; t_vm += t1_ - t_; n_bar++; } while (0)
#else
#define DUAL_BARRIER() __syncthreads()
#endif
constexpr uint32_t kDualR = 8192;
constexpr uint32_t kDualNear = kDualR - 2u * kT2; /* 5120 */
constexpr uint32_t kDualMaxBlocks = 3072;         /* 12 workgroups of two waves per CU (13 fit by the LDS, but 3 328 blocks run slower than on one wave: profiles/r04_lz4_dual_threshold.txt) */

struct DualLds {
    uint8_t ring[kDualR + 16];
    uint8_t in[kInRing + 16];
    uint32_t mmeta[64];
    uint32_t mbm[CopyLds<kDualR, kT2>::kWords];
    uint32_t da[2][64], db[2][64]; /* per sequence: ll | ml << 16;  off | ostart << 16 | far << 31 */
    uint32_t u[2][4];              /* op0, sequences, bytes, mode (0 nothing for B, 1 a batch, 2 the block is done) */
};

/* wave A: one batch up to and including its literal runs; publishes it in buffer `buf`.  Returns the sequences taken. */
__device__ inline uint32_t lz4_dual_front(Wave<kDualR> &w, DualLds &D, uint32_t &vp, const uint32_t B, const uint32_t epos,
                                          const uint32_t navail, const IndexRow<true> &ix, uint32_t &epre, const uint32_t buf)
{
    constexpr uint32_t R = kDualR;
    const uint32_t lane = w.lane;
    const uint32_t vend = w.vend;
    if (vend < 32u || B < 32u || navail == 0u) return 0;
    const uint32_t vsafe = vend - 16u;
    if (vp + 64u > vsafe || w.op + 64u > B) return 0;
    while (w.in_hi < vend && w.in_hi < vp + kSeqWin) w.refill_deferred();
    LDS_FENCE();
    uint2 mir = make_uint2(0, 0);
    if (lane < 2u) mir = *reinterpret_cast<const uint2 *>(D.in + lane * 8u);
    const uint32_t pos = vp + ((epos + w.delta - vp) & 0xffffu);
    const bool cand = lane < navail && pos + 8u <= vp + kSeqWin;
    const uint32_t rp = cand ? pos : vp;
    const uint32_t t = D.in[rp & kInMask];
    const uint32_t e1 = D.in[(rp + 1u) & kInMask];
    uint32_t ll = t >> 4;
    uint32_t k = 1u;
    if (ll == 15u) { ll += e1; k = 2u; }
    const uint32_t q = rp + k + ll;
    const bool inwin = q + 8u <= vp + kSeqWin;
    const uint32_t rq = inwin ? q : vp;
    const uint32_t off = (uint32_t)D.in[rq & kInMask] | ((uint32_t)D.in[(rq + 1u) & kInMask] << 8);
    const uint32_t e2 = D.in[(rq + 2u) & kInMask];
    if (lane < 2u) *reinterpret_cast<uint2 *>(D.in + kInRing + lane * 8u) = mir;
    uint32_t ml = (t & 15u) + 4u;
    uint32_t dlen = k + ll + 2u;
    const bool hasM = (t & 15u) == 15u;
    if (hasM) { ml += e2; dlen += 1u; }
    const uint32_t outlen = cand ? ll + ml : 0u;
    const uint32_t oend = scan64_incl(outlen);
    const uint32_t ostart = oend - outlen;
    const uint32_t mabs = w.op + ostart + ll;
    const uint32_t npos = lane_next(pos);
    const bool chain = (lane == 0u ? pos == vp : true) && (lane + 1u >= navail || lane == 63u || npos == pos + dlen);
    constexpr uint32_t kOvlMax = 64;
    const bool isfar = cand && off >= kDualNear;
    const bool ok = cand && inwin && chain && !(t >= 0xf0u && e1 == 255u) && !(hasM && e2 == 255u) && (off >= ml || (off != 0u && ml <= kOvlMax)) && off <= mabs &&
                    pos + dlen <= vsafe && oend <= kT2 && w.op + oend + 16u <= B && !(isfar && ml > 32u);
    const unsigned long long badmask = wave_ballot(!ok);
    const uint32_t nseq = badmask ? ctz64(badmask) : 64u;
    w.top_up();
    if (nseq == 0u) return 0;
    const uint32_t T = lane_get(oend, nseq - 1u);
    const uint32_t used = lane_get(pos + dlen, nseq - 1u) - vp;
    epre = ix.entry(nseq, lane);
    /* far matches are this wave's too: their sources are output it has flushed itself (everything below the batch the other
     * wave is working on), their destinations bytes of this batch; the trip to memory hides behind the literal runs */
    const bool farm = lane < nseq && isfar;
    uint4 xfa = make_uint4(0, 0, 0, 0), xfb = xfa;
    if (farm) {
        const uint8_t *g = w.dst + (mabs - off);
        __builtin_memcpy(&xfa, g, 16);
        __builtin_memcpy(&xfb, g + 16, 16);
    }
    /* the literal runs: nothing the other wave is doing can touch their destinations */
    {
        uint32_t spill = 0;
        const uint4 z = make_uint4(0, 0, 0, 0);
        lane_runs<R, kInMask>(D.ring, D.in, lane < nseq ? ll : 0u, pos + k, w.op + ostart, false, z, z, spill);
        lane_runs<R, R - 1u>(D.ring, D.ring, farm ? ml : 0u, 0u, mabs, true, xfa, xfb, spill);
        const unsigned long long sm = wave_ballot(spill != 0u);
        if (sm != 0ull) {
            LDS_FENCE();
            const uint32_t kk = lane_get(spill, ctz64(sm));
            if (lane < kk) D.ring[lane] = D.ring[R + lane];
        }
    }
    D.da[buf][lane] = ll | ((isfar ? 0u : ml) << 16); /* a far match is done: nothing of it for the other wave */
    D.db[buf][lane] = off | (ostart << 16);
    if (lane == 0u) { D.u[buf][0] = w.op; D.u[buf][1] = nseq; D.u[buf][2] = T; D.u[buf][3] = 1u; }
    w.op += T;
    vp += used;
    return nseq;
}

__global__ void __launch_bounds__(128)
k_lz4_dec_dual(const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
               const uint32_t *__restrict__ src_size, uint8_t *dst_base, uint64_t dst_stride, uint32_t B,
               uint64_t n_blocks, int32_t *__restrict__ status, const uint16_t *__restrict__ tbl, uint32_t tbl_cap,
               const uint2 *__restrict__ seg, const uint32_t logS, const uint32_t cap_s, const uint32_t ext,
               const uint32_t skip_heavy, const uint32_t *__restrict__ decoded, const uint64_t blk0)
{
    constexpr uint32_t R = kDualR;
    __shared__ __attribute__((aligned(16))) DualLds D;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t role = uni(threadIdx.x >> 6); /* 0: wave A, 1: wave B */
    const uint64_t blk = blk0 + blockIdx.x;
    if (blk >= n_blocks) return;
    const uint8_t *base = src_base + uni64(src_off[blk]);
    const uint32_t csize = uni(src_size[blk]);
    if (skip_heavy != 0u && lz4_literal_heavy(csize, B)) return;
    if (decoded != nullptr && uni(decoded[blk]) != 0u) return;

    Wave<R> w;
    w.ring = D.ring;
    w.in = D.in;
    w.lane = lane;
    w.delta = (uint32_t)(reinterpret_cast<uintptr_t>(base) & 127u);
    w.abase = base - w.delta;
    w.vend = w.delta + csize;
    w.in_hi = 0;
    w.dst = dst_base + uni64(blk * dst_stride);
    w.dst_aligned = (reinterpret_cast<uintptr_t>(w.dst) & 15u) == 0;
    w.op = 0;
    w.flushed = 0;
    w.pre = w.pre2 = w.pre3 = make_uint2(0, 0);
    w.nstale = 0;
    Stats st = {};
#if CRYO_DUAL_PROF
    unsigned long long t_wait = 0, t_vm = 0, n_bar = 0;
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif

    if (role == 1u) {
        /* ---- wave B: the output-ring half of every batch ---- */
        const CopyLds<R, kT2> L = {D.ring, D.in, D.mmeta, D.mbm};
        uint32_t buf = 0;
        for (;;) {
            DUAL_BARRIER();
            const uint32_t mode = uni(D.u[buf][3]);
            if (mode == 2u) break;
            if (mode == 1u) {
                const uint32_t op0 = uni(D.u[buf][0]), nseq = uni(D.u[buf][1]), T = uni(D.u[buf][2]);
                const uint32_t a = D.da[buf][lane], b = D.db[buf][lane];
                const uint32_t ll = a & 0xffffu, ml = a >> 16, off = b & 0xffffu, ostart = b >> 16;
                const bool act = lane < nseq;
                /* the ring's first bytes behind its end, for a 16-byte source read that starts in its last 15 -- only when a
                 * source run of this batch crosses the end: that end lies below the batch, more than R - 2T behind it, so
                 * neither this batch nor the other wave's crosses the ring's end and the 16 bytes are nobody's landing strip.
                 * (A read that crosses without its run doing so fetches bytes nobody uses.) */
                const uint32_t mrel = ostart + ll;
                const bool near_indep = act && ml != 0u && off >= mrel + ml;
                if (wave_any(near_indep && ((op0 + mrel - off) & (R - 1u)) + ml > R)) {
                    uint2 mir = make_uint2(0, 0);
                    if (lane < 2u) mir = *reinterpret_cast<const uint2 *>(D.ring + lane * 8u);
                    LDS_FENCE();
                    if (lane < 2u) *reinterpret_cast<uint2 *>(D.ring + R + lane * 8u) = mir;
                    LDS_FENCE();
                }
                w.op = op0;
                const uint4 z = make_uint4(0, 0, 0, 0);
                seq_copy<R, kT2, true>(w, L, nseq, ostart, ll, ml, off, 0u, T, false, z, z, st);
            }
            buf ^= 1u;
        }
#if CRYO_DUAL_PROF
        if (blk == 0 && lane == 0) printf("[dual] B: %llu barriers, memory %llu, waited %llu of %llu ticks\n", n_bar, t_vm, t_wait, __builtin_amdgcn_s_memtime() - t_start);
#endif
        return;
    }

    /* ---- wave A ---- */
    uint32_t vp = w.delta;
    bool bad = (csize == 0);
    bool done = bad;
    uint32_t skip = 0;
    IndexRow<true> ix;
    ix.open(tbl + uni64(blk * (uint64_t)tbl_cap), seg, blk, logS, cap_s, ext, lane);
    uint32_t poor = 0;
    uint32_t efirst = 0;
    if (!bad) efirst = ix.entry(0, lane);
    bool have_first = !bad;
    if (!bad) {
        w.prefetch();
        w.refill();
        if (w.in_hi < w.vend) w.refill();
    }
    uint32_t npre = ix.avail(0);
    uint32_t buf = 0;
    /* a step = publish + barrier: B starts on what was published, and has finished the step before */
    auto step_batch = [&](const uint32_t op0) {
        DUAL_BARRIER();
        buf ^= 1u;
        /* everything below the batch B is starting on is complete: store whole chunks of it */
        const uint32_t op_now = w.op;
        w.op = op0;
        w.flush();
        w.op = op_now;
    };
    auto park_b = [&]() { /* nothing for B: it reads the mode and waits at the next barrier; the ring is A's until then */
        if (lane == 0u) D.u[buf][3] = 0u;
        DUAL_BARRIER();
        buf ^= 1u;
    };

    while (!done) {
        if (skip == 0u) {
            uint32_t n;
            uint32_t epre = efirst;
            bool have_pre = have_first;
            bool more;
            have_first = false;
            do {
                uint32_t e = epre, navail = npre;
                if (!have_pre) { navail = ix.avail(0); e = ix.entry(0, lane); }
                const uint32_t op0 = w.op;
                n = lz4_dual_front(w, D, vp, B, e, navail, ix, epre, buf);
                if (n != 0u) step_batch(op0);
                have_pre = n != 0u;
                npre = ix.avail(n);
                ix.advance(n);
                more = n >= 8u || (n != 0u && n == navail);
            } while (more);
            poor = n < 4u ? poor + 1u : 0u;
            if (poor >= 3u) { skip = 8u; poor = 0u; }
        } else {
            skip--;
        }
        ix.advance(1u);
        park_b(); /* B has finished the batch before; the general path owns the ring */
        const uint32_t r = lz4_general_seq<R>(w, vp, B);
        if (r == 2u) { bad = true; break; }
        if (r == 1u) done = true;
    }
    if (lane == 0u) D.u[buf][3] = 2u;
    DUAL_BARRIER();
    if (!bad && w.op != B) bad = true;
    if (!bad) {
        w.flush();
        w.flush_tail();
    }
    if (lane == 0) status[blk] = bad ? CRYO_ST_CORRUPT : CRYO_ST_OK;
#if CRYO_DUAL_PROF
    if (blk == 0 && lane == 0) printf("[dual] A: %llu barriers, memory %llu, waited %llu of %llu ticks\n", n_bar, t_vm, t_wait, __builtin_amdgcn_s_memtime() - t_start);
#endif
}

/* ---- launcher ---- */
/* blocks one round of k_lz4_dec_seq holds: 24 waves per CU (LDS, 6 720 bytes per wave) on the device's CUs (256 on an
 * MI355X; a partitioned device has fewer: Lz4DecodeOpts::cus, filled by the handle from hipDeviceProp) */
static hipError_t launch_dec_seq(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off, const uint32_t *d_src_size, uint8_t *d_dst,
                           uint64_t dst_stride, uint32_t block_size, uint64_t n_blocks, int32_t *d_status, const void *ws,
                           const Lz4IndexLayout &Lx, const uint32_t *d_done = nullptr, const int waves = 0, const Lz4DecodeOpts *opts = nullptr)
{
    const uint16_t *tbl = static_cast<const uint16_t *>(ws);
    const uint2 *seg = reinterpret_cast<const uint2 *>(static_cast<const uint8_t *>(ws) + Lx.seg_off);
    const uint32_t heavy = Lx.logS != 0u ? 1u : 0u;
    auto dual = [&](hipStream_t q, uint64_t first) {
        hipLaunchKernelGGL(k_lz4_dec_dual, dim3((uint32_t)(n_blocks - first)), dim3(128), 0, q, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size,
                           n_blocks, d_status, tbl, Lx.cap, seg, Lx.logS, Lx.cap_main + Lx.ext, Lx.ext, heavy, d_done, first);
    };
    if (waves == 2 || (waves == 0 && n_blocks <= kDualMaxBlocks)) { /* a batch that leaves most of the chip idle: two waves per block */
        dual(s, 0);
        return hipGetLastError();
    }
    /* The last round (round 5).  A block is its wave's chain, so a batch of 1.33 rounds -- 8 192 blocks of 1 MiB, the reference's
     * block size -- takes two rounds' time: the last 2 048 blocks run on a chip that is two thirds empty at the pace of a full
     * one.  Those blocks go to a low-priority side stream with two waves each (1.4-1.7 x per block, k_lz4_dec_dual): its
     * workgroups are placed as the first part's retire.  Measured (profiles/r05_lz4_tail_round.txt): 7 168 x 1 MiB 8.26 -> 7.50 ms,
     * 8 192 x 1 MiB 8.58 -> 8.13, 7 000 x 128 KiB 1.25 -> 1.16; with more than two rounds, or a last round of more than a third,
     * it loses 1-6 % (the two-wave workgroups wait for LDS the one-wave ones hold): only between one and two rounds, and for
     * a last round of at most 2 048 blocks. */
    uint64_t head = n_blocks;
    if (waves == 0 && opts != nullptr && opts->side != nullptr && opts->fork != nullptr && opts->join != nullptr) {
        const uint64_t resident = (uint64_t)(opts->cus > 0 ? opts->cus : 256) * 6u * kDecWpb;
        if (n_blocks > resident && n_blocks <= 2u * resident) {
            const uint64_t rest = n_blocks % resident;
            if (rest != 0u && rest <= 2048u) head = n_blocks - rest;
        }
    }
    if (head != n_blocks) { /* behind the index pass; a fork that cannot be made means one stream for the whole batch */
        if (hipEventRecord(opts->fork, s) != hipSuccess || hipStreamWaitEvent(opts->side, opts->fork, 0) != hipSuccess) {
            (void)hipGetLastError();
            head = n_blocks;
        }
    }
    const dim3 g((uint32_t)((head + kDecWpb - 1) / kDecWpb)), b(64 * kDecWpb);
    /* (the first part's kernel is told where its blocks end by its grid: a workgroup's four blocks never straddle `head`,
     * a multiple of the residency) */
    if (Lx.logS != 0u)
        hipLaunchKernelGGL((k_lz4_dec_seq<kDecR, false, kDecWpb, true>), g, b, 0, s, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size, head,
                           d_status, nullptr, tbl, Lx.cap, seg, Lx.logS, Lx.cap_main + Lx.ext, Lx.ext, 1u, d_done, (uint64_t)0);
    else
        hipLaunchKernelGGL((k_lz4_dec_seq<kDecR, false, kDecWpb, false>), g, b, 0, s, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size, head,
                           d_status, nullptr, tbl, Lx.cap, seg, Lx.logS, Lx.cap_main + Lx.ext, Lx.ext, 0u, d_done, (uint64_t)0);
    if (head != n_blocks) {
        dual(opts->side, head);
        /* the caller's stream must not run on (status copy, timer, the next call) before the side stream's blocks are done */
        if (hipError_t e = hipEventRecord(opts->join, opts->side); e != hipSuccess) { (void)hipStreamSynchronize(opts->side); return e; }
        if (hipError_t e = hipStreamWaitEvent(s, opts->join, 0); e != hipSuccess) { (void)hipStreamSynchronize(opts->side); return e; }
    }
    return hipGetLastError();
}

/* the blocks the few-blocks path did not decode (lz4_lat.hip), with the index it built */
hipError_t launch_lz4_dec_seq_rest(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off, const uint32_t *d_src_size,
                                   uint8_t *d_dst, uint64_t dst_stride, uint32_t block_size, uint64_t n_blocks, int32_t *d_status,
                                   const void *ws, const Lz4IndexLayout &Lx, const uint32_t *d_done)
{
    return launch_dec_seq(s, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size, n_blocks, d_status, ws, Lx, d_done);
}

hipError_t launch_lz4_decompress_indexed(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                                         const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                                         uint32_t block_size, uint64_t n_blocks, int32_t *d_status, void *d_workspace,
                                         size_t workspace_bytes, uint32_t walkers, int waves, const Lz4DecodeOpts *opts)
{
    if (n_blocks == 0) return hipSuccess;
    const uint64_t grid = (n_blocks + kDecWpb - 1) / kDecWpb;
    if (grid > 0x7fffffffull || !d_workspace) return hipErrorInvalidValue;
    const Lz4IndexLayout Lx = lz4_index_layout(n_blocks, block_size, walkers);
    if (workspace_bytes < Lx.bytes) return hipErrorInvalidValue;
    if (hipError_t e = launch_lz4_index(s, d_src, d_src_off, d_src_size, n_blocks, block_size, d_workspace, Lx); e != hipSuccess) return e;
    const dim3 g((uint32_t)grid), b(64 * kDecWpb);
#ifdef CRYO_DEBUG
    /* phase timing and ablation of the decoder (profiles/scripts): a debug build only -- an ablated run decodes wrong bytes */
    static const bool want_stats = cryo_tuning_env("CRYO_LZ4_STATS") != nullptr;
    if (want_stats) {
        unsigned long long *d_st = nullptr, h_st[32];
        if (hipMalloc((void **)&d_st, sizeof h_st) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d_st, 0, sizeof h_st, s);
        static const unsigned long long abl = cryo_tuning_env("CRYO_LZ4_ABLATE") ? strtoull(cryo_tuning_env("CRYO_LZ4_ABLATE"), nullptr, 0) : 0ull;
        (void)hipMemcpyAsync(d_st + 7, &abl, sizeof abl, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL((k_lz4_dec_seq<kDecR, true, kDecWpb, true>), g, b, 0, s, d_src, d_src_off, d_src_size, d_dst, dst_stride,
                           block_size, n_blocks, d_status, d_st, static_cast<const uint16_t *>(d_workspace), Lx.cap,
                           reinterpret_cast<const uint2 *>(static_cast<const uint8_t *>(d_workspace) + Lx.seg_off), Lx.logS, Lx.cap_main + Lx.ext, Lx.ext, 0u, nullptr, (uint64_t)0);
        (void)hipMemcpyAsync(h_st, d_st, sizeof h_st, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        (void)hipFree(d_st);
        fprintf(stderr, "[lz4 seq stats] batches %llu batch_seqs %llu general_seqs %llu match chunks %llu rounds %llu zero_batches %llu\n",
                h_st[0], h_st[1], h_st[2], h_st[3], h_st[4], h_st[5]);
        unsigned long long tot = 0;
        for (int k = 0; k < 8; k++) tot += h_st[8 + k];
        static const char *nm[8] = {"stage", "decode+validate", "flush", "requests", "bitmap+meta", "lane runs", "chunk bases+match space", "general+other"};
        for (int k = 0; k < 8; k++) fprintf(stderr, "[lz4 seq cycles] %-16s %5.1f%%\n", nm[k], 100.0 * (double)h_st[8 + k] / (double)(tot ? tot : 1));
        fprintf(stderr, "[lz4 seq stops] not a candidate %llu, offset outside the window %llu, chain %llu, literal 255-run %llu, match 255-run %llu, "
                        "overlapping match %llu, offset too far %llu, end of input %llu, batch full (T) %llu, end of output %llu, far and long %llu\n",
                h_st[16], h_st[17], h_st[18], h_st[19], h_st[20], h_st[21], h_st[22], h_st[23], h_st[24], h_st[25], h_st[26]);
        return hipGetLastError();
    }
#endif
    (void)g; (void)b;
    if (hipError_t e = launch_dec_seq(s, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size, n_blocks, d_status, d_workspace, Lx, nullptr, waves, opts);
        e != hipSuccess)
        return e;
    if (Lx.logS != 0u) /* the blocks the index and its decoder left out (almost all literals): the in-wave parser */
        return launch_lz4_dec_ring(s, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size, n_blocks, d_status, true);
    return hipGetLastError();
}

} // namespace cryo
