/*
 * lz4_dec2.hip -- LZ4 block decode for large batches: sequence index pass + run-space copy engine.
 *
 * Replaces LZ4_decompress_safe(compressed, out, compressed_size, CRYO_BLCKSZ) (reference
 * compression.c:84) for a batch of independent blocks, with the accept/reject rules listed at the top of
 * lz4_dec.hip (whose kernel stays the path for small batches).  Two kernels:
 *
 *   k_lz4_index   one LANE per block: walks the token chain and writes the low 16 bits of every token
 *                 position to the block's row of the workspace (the serial part of LZ4 decoding, run
 *                 for all blocks of the batch at once);
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
 * Sequence index.  Finding where the sequences of an LZ4 block start is a serial walk (token ->
 * literal length -> next token); done inside the decoding wave it needs speculative per-byte tables
 * (lz4_dec.hip, lz4_batch: ~7 VALU instructions and four dependent LDS passes per sequence).  Across a
 * batch the walk is embarrassingly parallel.
 *
 * k_lz4_index: 64 blocks per wave (every lane walks one block's token chain), 4 waves per CU (8 with the
 * smaller rings), so every block of a 64k-block batch has its walk in flight at once and the pass takes
 * (sequences per block) x (one hop).
 *
 * A lane that read its block straight from global memory paid ~1 us per hop (64 lanes = 64 cache lines
 * per load, every hop a dependent trip to L2 or beyond: 6.9 ms for the headline batch, measured).  So each
 * block's stream is staged through a private 512-byte LDS ring, filled 128 bytes at a time: in turn j of
 * four the wave's 64 lanes load one chunk for each of blocks 8j..8j+7 (8 lanes x 16 bytes per block, one
 * cache line) if that block has room, and store it into the ring one round of turns later, so the load's
 * latency is covered by four hops.  The walk is a small state machine per lane (token / literal-length
 * extension / match-length extension) so that one LDS read per hop serves every lane, whatever it is in.
 * --------------------------------------------------------------------------------------------- */
constexpr uint32_t kIdxLanes = 64, kIdxChunk = 128;

__device__ inline uint32_t bperm(uint32_t v, uint32_t src_lane)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}

template <uint32_t kIdxRing> /* bytes of LDS ring per block: 512, or 256 (twice the waves per CU) */
__global__ void __launch_bounds__(64)
k_lz4_index(const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
            const uint32_t *__restrict__ src_size, uint64_t n_blocks, uint16_t *__restrict__ tbl, uint32_t cap,
            uint32_t *__restrict__ tbl_n)
{
    constexpr uint32_t kIdxStride = kIdxRing + 16u; /* bank skew between rings */
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[kIdxLanes * kIdxStride];
    /* the last 16 positions of every lane (4 groups of 4) + one slot where a lane that records nothing writes */
    __shared__ __attribute__((aligned(8))) uint16_t s_pos[kIdxLanes][20];
    const uint32_t lane = threadIdx.x;
    const uint64_t blk = (uint64_t)blockIdx.x * kIdxLanes + (lane & (kIdxLanes - 1u));
    const bool owner = blk < n_blocks;
    /* stream of this lane's block, in "virtual" positions: vp = delta + offset in the block, so that chunk
     * addresses are 128-byte aligned */
    uint64_t aoff = 0;
    uint32_t delta = 0, vend = 0;
    if (owner) {
        const uint64_t o = src_off[blk];
        aoff = o & ~(uint64_t)127;     /* chunks are whole 128-byte lines: each line of the input is fetched once */
        delta = (uint32_t)(o & 127u);
        vend = delta + src_size[blk];
    }
    uint16_t *row = tbl + blk * cap;
    uint16_t *dummy = tbl + n_blocks * cap + lane * 8u; /* 8 bytes per lane behind the rows: where lanes without a block store */
    if (!owner) aoff = src_off[0] & ~(uint64_t)127; /* a lane past the end of the batch re-reads block 0 */
    uint32_t pos = delta;        /* next byte to interpret */
    uint32_t requested = 0;      /* chunks requested up to here (multiple of kIdxChunk) */
    uint32_t filled = 0;         /* chunks stored in the ring up to here */
    uint32_t outst = 0, drop = 0; /* chunks of this lane on their way; how many of them a restart of the ring disowned */
    uint32_t state = 0;          /* 0 token, 1 literal-length extension, 2 match-length extension */
    uint32_t acc = 0, tm = 0;    /* literal length being accumulated; match nibble of the current token */
    uint32_t k = 0;
    uint16_t *pbuf = s_pos[lane & (kIdxLanes - 1u)];
    bool done = !owner || vend == delta;

    /* what this lane serves in turn j: one 16-byte piece of the next chunk of blocks 16j + (lane >> 3) and
     * 16j + 8 + (lane >> 3) (two loads per turn: every block has a turn every fourth hop) */
    const uint32_t piece16 = (lane & 7u) * 16u;
#define IDX_SRC(j) const uint64_t saoff##j = ((uint64_t)bperm((uint32_t)(aoff >> 32), 8u * j + (lane >> 3)) << 32) | bperm((uint32_t)aoff, 8u * j + (lane >> 3)); \
                   const uint32_t svend##j = bperm(vend, 8u * j + (lane >> 3));
    IDX_SRC(0) IDX_SRC(1) IDX_SRC(2) IDX_SRC(3) IDX_SRC(4) IDX_SRC(5) IDX_SRC(6) IDX_SRC(7)
#undef IDX_SRC
    const uint32_t rb = (lane & (kIdxLanes - 1u)) * kIdxStride; /* this lane's ring inside s_ring */

    /* chunks on their way: two per turn, committed TWO rounds later (a lane with room in its ring requests one chunk
     * per round, up to two outstanding; four rounds of distance bought nothing and its 32 slots spilled registers).
     * Separate variables, not arrays: the compiler kept an indexed array in scratch memory. */
#define IDX_SLOT(n) uint4 fd##n = make_uint4(0, 0, 0, 0), fe##n = fd##n; uint32_t fa##n = 0, fb##n = 0; bool fp##n = false, fq##n = false, fo##n = false;
    IDX_SLOT(0) IDX_SLOT(1) IDX_SLOT(2) IDX_SLOT(3) IDX_SLOT(4) IDX_SLOT(5) IDX_SLOT(6) IDX_SLOT(7)
#undef IDX_SLOT

    auto turn = [&](const uint32_t j, uint4 &fd, uint4 &fe, uint32_t &fa, uint32_t &fb, bool &fp, bool &fq, bool &fpo,
                    const uint64_t soff, const uint32_t sve, const uint64_t soff2, const uint32_t sve2) __attribute__((always_inline)) {
        const bool myturn = (lane >> 4) == j; /* lanes 16j..16j+15 */
        /* ---- commit the chunks requested four rounds ago ---- */
        if (fp) *reinterpret_cast<uint4 *>(s_ring + fa) = fd;
        if (fq) *reinterpret_cast<uint4 *>(s_ring + fb) = fe;
        if (myturn && outst != 0u && fpo) { /* fpo: this lane did request in the turn being committed */
            outst--;
            if (drop != 0u) drop--; else filled += kIdxChunk;
        }
        /* the hop's two ring reads go out before the exchange below: one LDS round trip per turn, not two */
        const uint32_t w0 = *reinterpret_cast<const uint32_t *>(s_ring + rb + (pos & (kIdxRing - 4u)));
        const uint32_t w1 = *reinterpret_cast<const uint32_t *>(s_ring + rb + ((pos + 4u) & (kIdxRing - 4u)));
        const uint32_t w2 = *reinterpret_cast<const uint32_t *>(s_ring + rb + ((pos + 8u) & (kIdxRing - 4u)));
        /* ---- request the next chunk of blocks 8j..8j+7 (one bpermute: requested | want) ---- */
        {
            const bool want = myturn && !done && requested < vend && pos + (kIdxRing - kIdxChunk) >= requested;
            const uint32_t msg = requested | (want ? 1u : 0u);
            fpo = want;
            if (want) { requested += kIdxChunk; outst++; }
            const uint32_t s1 = 16u * j + (lane >> 3), s2 = s1 + 8u;
            const uint32_t m1 = bperm(msg, s1), m2 = bperm(msg, s2);
            const uint32_t o1 = (m1 & ~1u) + piece16, o2 = (m2 & ~1u) + piece16;
            fp = (m1 & 1u) != 0u;
            fq = (m2 & 1u) != 0u;
            fa = s1 * kIdxStride + (o1 & (kIdxRing - 1u));
            fb = s2 * kIdxStride + (o2 & (kIdxRing - 1u));
            /* always two loads per turn (a lane with nothing to fetch re-reads its block's first 16 bytes): with a
             * fixed number of vector-memory operations per turn the compiler can wait for exactly the chunks it
             * commits (vmcnt(N)); a conditional load made it drain the queue once per round (2.3 us a round) */
            fd = *reinterpret_cast<const uint4 *>(src_base + (soff + ((fp && o1 < sve) ? o1 : 0u)));
            fe = *reinterpret_cast<const uint4 *>(src_base + (soff2 + ((fq && o2 < sve2) ? o2 : 0u)));
        }
        /* ---- one hop, branch-free for the two common states (token, match-length extension) ---- */
        {
            const bool live = !done && pos < vend;
            const bool canread = pos < requested && (pos + 8u <= filled || filled >= vend);
            const uint32_t x = __builtin_amdgcn_alignbyte(w1, w0, pos & 3u);
            const bool go = live && canread && state != 1u;
            /* token */
            const uint32_t ll = (x >> 4) & 15u, e1 = (x >> 8) & 255u, tmn = x & 15u;
            const bool l15 = ll == 15u;
            const uint32_t q2 = pos + 3u + ll + (l15 ? e1 + 1u : 0u);  /* behind the literals and the offset */
            const bool tok = go && state == 0u;
            const bool longlit = tok && l15 && e1 == 255u;            /* 255-run: slow path below */
            /* match-length extension bytes */
            const uint32_t nx = ~x;
            const uint32_t n = nx ? (uint32_t)__builtin_ctz(nx) >> 3 : 4u; /* leading 0xFF bytes */
            const uint32_t adv = n == 4u ? 4u : n + 1u;
            const bool ext = go && state == 2u;
            /* record + advance */
            const bool rec = tok;
            pbuf[rec ? (k & 15u) : 16u] = (uint16_t)(pos - delta); /* unconditional: a store in a branch costs more than the branch saves */
            if (rec) { k++; tm = tmn; }
            const bool fin = tok && !longlit && q2 > vend;              /* last sequence: literals only */
            if (tok && !longlit && !fin) { pos = q2; state = tmn == 15u ? 2u : 0u; }
            /* a second token in the same turn when the first one leaves it inside the eight bytes just read: no or
             * up to two literals and a short match (half of the sequences of tuple data) */
            {
                const bool dbl = tok && !longlit && !fin && !l15 && tmn != 15u && ll <= 2u && q2 < vend && k + 1u < cap;
                const uint32_t x1 = __builtin_amdgcn_alignbyte(w2, w1, (q2 - 3u - ll) & 3u); /* bytes 4..7 behind the first token */
                const unsigned long long xx = ((unsigned long long)x1 << 32) | x;
                const uint32_t y = (uint32_t)(xx >> (8u * (3u + ll)));
                const uint32_t llb = (y >> 4) & 15u, e1b = (y >> 8) & 255u, tmb = y & 15u;
                const bool l15b = llb == 15u;
                const uint32_t q2b = q2 + 3u + llb + (l15b ? e1b + 1u : 0u);
                const bool rec2 = dbl && !(l15b && e1b == 255u);
                pbuf[rec2 ? (k & 15u) : 16u] = (uint16_t)(q2 - delta);
                if (rec2) {
                    k++;
                    tm = tmb;
                    if (q2b > vend) done = true;
                    else { pos = q2b; state = tmb == 15u ? 2u : 0u; }
                }
            }
            if (ext) { pos += adv; state = n == 4u ? 2u : 0u; }
            if (longlit) { state = 1u; acc = 15u + 255u; pos += 2u; }
            if (fin || (!done && !live) || k >= cap) done = true;
            /* rare: literal-length 255-runs, and jumps over everything requested (a long literal run) */
            const bool slow = !done && (state == 1u || pos >= requested) && !longlit;
            if (__any(slow)) {
                if (slow && pos < vend) {
                    if (pos >= requested) {
                        /* restart the ring at the chunk of pos; a chunk still in flight lands in a slot that is
                         * rewritten before it is read */
                        requested = filled = pos & ~(kIdxChunk - 1u);
                        drop = outst;
                    } else if (state == 1u && live && canread) {
                        if (n == 4u) { acc += 1020u; pos += 4u; if (acc >= vend) done = true; }
                        else {
                            acc += 255u * n + ((x >> (8u * n)) & 255u);
                            const uint32_t q = pos + n + 1u + acc;
                            if (acc >= vend || q + 2u > vend) done = true;
                            else { pos = q + 2u; state = tm == 15u ? 2u : 0u; }
                        }
                    }
                }
            }
        }
    };

    /* positions go out in aligned groups of four (8 bytes): once per round the group being filled and the two before
     * it (a lane gains at most eight positions per round, so no group is missed); a group is stored a few times while
     * it fills, always whole and aligned, so the L2 merges the stores of a row into full lines.  (Storing "the last
     * four entries" at a 2-byte granular address instead wrote 5.2 GB for a 0.8 GB index.)  Three unconditional
     * stores per round, see the note on the loads.  A macro, not a lambda: captured by a lambda, the packs lived in
     * scratch memory. */
#define IDX_PUT()                                                                                            \
    {                                                                                                        \
        const uint32_t g = k ? (k - 1u) >> 2 : 0u;                                                           \
        const uint32_t g2 = g > 1u ? g - 2u : 0u, g1 = g ? g - 1u : 0u;                                      \
        uint16_t *r = owner ? row : dummy;                                                                   \
        unsigned long long v2, v1, v0;                                                                       \
        __builtin_memcpy(&v2, pbuf + 4u * (g2 & 3u), 8);                                                     \
        __builtin_memcpy(&v1, pbuf + 4u * (g1 & 3u), 8);                                                     \
        __builtin_memcpy(&v0, pbuf + 4u * (g & 3u), 8);                                                      \
        __builtin_memcpy(r + 4u * g2, &v2, 8);                                                               \
        __builtin_memcpy(r + 4u * g1, &v1, 8);                                                               \
        __builtin_memcpy(r + 4u * g, &v0, 8);                                                                \
    }
#define IDX_TURN(j, n, sa, sb) turn(j, fd##n, fe##n, fa##n, fb##n, fp##n, fq##n, fo##n, saoff##sa, svend##sa, saoff##sb, svend##sb);
#define IDX_ROUND(a, b, c, d)                                   \
    IDX_PUT()                                                   \
    IDX_TURN(0, a, 0, 1) IDX_TURN(1, b, 2, 3) IDX_TURN(2, c, 4, 5) IDX_TURN(3, d, 6, 7)
    while (__any(!done)) {
        IDX_ROUND(0, 1, 2, 3)
        IDX_ROUND(4, 5, 6, 7)
    }
#undef IDX_ROUND
#undef IDX_TURN
    if (owner) {
        IDX_PUT()
        tbl_n[blk] = k;
    }
}

/* ---------------------------------------------------------------------------------------------
 * Decoder
 * --------------------------------------------------------------------------------------------- */

constexpr uint32_t kSeqWin = 1536;  /* compressed bytes a batch may span (in_hi stays < vp + kInRing)   */
constexpr uint32_t kT2 = 1536;      /* output bytes per batch: the largest T with R - T >= T + 1023 at R = 4096, i.e. every
                                     * offset is either still in the ring (off < R - T) or flushed (off >= T + 1023) */

/* lane i gets lane i+1's value (lane 63: 0) */
__device__ inline uint32_t lane_next(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

/*
 * One batch: sequences n0 .. n0+63 of the block start at the positions in the index row (epos = this lane's
 * entry).  Returns the number of sequences decoded (0: the caller takes one sequence through the general path).
 */
template <uint32_t R>
__device__ inline uint32_t lz4_seq_batch(Wave<R> &w, const CopyLds<R, kT2> &L, uint32_t &vp, const uint32_t B,
                                         const uint32_t epos, const uint32_t navail, const uint16_t *__restrict__ trow,
                                         const uint32_t n0, const uint32_t ntab, uint32_t &epre, Stats &st,
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
    const unsigned long long badmask = __ballot(!ok);
    const uint32_t nseq = badmask ? ctz64(badmask) : 64u;
    if (st.on && stop_hist && badmask && nseq < navail && lane == nseq) { /* why the batch stops in front of this sequence */
        const int why = !cand ? 0 : !inwin ? 1 : !chain ? 2 : (t >= 0xf0u && e1 == 255u) ? 3 : (hasM && e2 == 255u) ? 4 : (off < ml && (off == 0u || ml > kOvlMax)) ? 5 : off > mabs ? 6
                        : pos + dlen > vsafe ? 7 : oend > kT2 ? 8 : w.op + oend + 16u > B ? 9 : 10;
        atomicAdd(&stop_hist[why], 1ull);
    }
    stamp(st, 1);
    if (!(st.ablate & 2u) && !(CRYO_ABL & 64)) w.flush();   /* what earlier batches produced; far sources below are read back from it */
    else if (CRYO_ABL & 64) w.flushed = w.op & ~(kChunk - 1u);
    else w.flushed = w.op & ~(kChunk - 1u);
    stamp(st, 2);
    if (!(st.ablate & 4u)) w.top_up(); else w.nstale = 0;
    if (nseq == 0u) return 0;
    const uint32_t T = lane_get(oend, nseq - 1u);
    const uint32_t used = lane_get(pos + dlen, nseq - 1u) - vp;
    /* the next batch's positions are requested now: their trip to memory hides behind this batch's copy */
    epre = 0;
    if (n0 + nseq + lane < ntab) epre = trow[n0 + nseq + lane];
    /* ... and so are the sources of its far matches (flushed output: off >= T + 1023 behind a match) */
    uint4 xfa = make_uint4(0, 0, 0, 0), xfb = xfa;
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
            const unsigned long long m = __ballot(win != 255u) & ~((1ull << k) - 1ull);
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
            const unsigned long long m = (k < 64u) ? (__ballot(win != 255u) & ~((1ull << k) - 1ull)) : 0ull;
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

template <uint32_t R, bool STATS>
__global__ void __launch_bounds__(256, 6)
k_lz4_dec_seq(const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
              const uint32_t *__restrict__ src_size, uint8_t *dst_base, uint64_t dst_stride, uint32_t B,
              uint64_t n_blocks, int32_t *__restrict__ status, unsigned long long *stats,
              const uint16_t *__restrict__ tbl, uint32_t tbl_cap, const uint32_t *__restrict__ tbl_n)
{
    Stats st = {};
    st.on = STATS;
    if (STATS) { st.ablate = (uint32_t)stats[7]; st.t0 = __builtin_amdgcn_s_memtime(); }
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[4][R + 16];
    __shared__ __attribute__((aligned(16))) uint8_t s_in[4][kInRing + 16];
    __shared__ __attribute__((aligned(8))) uint32_t s_mmeta[4][64]; /* packed: 26 880 bytes per workgroup, six per CU */
    __shared__ __attribute__((aligned(8))) uint32_t s_mbm[4][CopyLds<R, kT2>::kWords];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = uni(threadIdx.x >> 6);
    const uint64_t blk = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid;
    if (blk >= n_blocks) return;

    const uint8_t *base = src_base + uni64(src_off[blk]);
    const uint32_t csize = uni(src_size[blk]);

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
    const uint16_t *trow = tbl + uni64(blk * (uint64_t)tbl_cap);
    const uint32_t ntab = uni(tbl_n[blk]);
    uint32_t n0 = 0; /* sequences decoded so far */
    uint32_t poor = 0;

    /* the first batch's index entries travel with the first input chunks (one trip to memory at the start of a block, not two) */
    uint32_t efirst = 0;
    if (!bad && lane < ntab) efirst = trow[lane];
    bool have_first = !bad;
    if (!bad) {
        w.prefetch();
        w.refill();
        if (w.in_hi < w.vend) w.refill();
    }

    while (!done) {
        if (skip == 0u) {
            uint32_t n;
            uint32_t epre = efirst;
            bool have_pre = have_first;
            have_first = false;
            do {
                uint32_t e = epre;
                const uint32_t navail = n0 < ntab ? (ntab - n0 < 64u ? ntab - n0 : 64u) : 0u;
                if (!have_pre) { e = 0; if (lane < navail) e = trow[n0 + lane]; }
                n = lz4_seq_batch<R>(w, L, vp, B, e, navail, trow, n0, ntab, epre, st, STATS ? stats + 16 : nullptr);
                have_pre = n != 0u;
                n0 += n;
                if (n == 0u) st.zero_batches++;
            } while (n >= 8u);
            /* a batch stops in front of a sequence it cannot take (overlapping match, long run, end of block):
             * that one goes through the general path and the batches resume.  Only data that keeps yielding
             * short batches (runs of overlapping matches) stays on the general path for a while. */
            poor = n < 4u ? poor + 1u : 0u;
            if (poor >= 3u) { skip = 8u; poor = 0u; }
        } else {
            skip--;
        }
        st.general_seqs++;
        n0++;
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

/* ---- launcher ---- */
/* Entries per row: B/8 + 64 rounded up to 1024 entries, i.e. rows 2 KiB-granular (34 816 bytes at 128 KiB blocks).  The
 * index pass writes 64 rows at once, 8 bytes each, and is sensitive to the row stride: with 33 024 bytes (+64 entries)
 * it always runs 17 % slower (3.6 instead of 3.05 ms), with the natural 32 896 sometimes, with 33 792 / 34 816 never
 * in 30 samples each (profiles/scripts/r02_cap.sh). */
uint32_t lz4_index_cap(uint32_t block_size)
{
    static const uint32_t pad = getenv("CRYO_LZ4_IDX_CAP_PAD") ? (uint32_t)atoi(getenv("CRYO_LZ4_IDX_CAP_PAD")) & ~3u : 0u; /* layout experiments */
    return ((block_size / 8u + 64u + 1023u) & ~1023u) + pad;
}

size_t lz4_index_workspace(uint64_t n_blocks, uint32_t block_size)
{
    return (size_t)n_blocks * lz4_index_cap(block_size) * 2u + 1024u /* dummy slots */ + (size_t)n_blocks * 4u + 64u;
}

hipError_t launch_lz4_index(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off, const uint32_t *d_src_size,
                            uint64_t n_blocks, uint16_t *tbl, uint32_t cap, uint32_t *tbl_n)
{
    /* 512-byte rings admit four waves per CU (one per SIMD).  256-byte rings (eight waves per CU) were measured and lose:
     * 65 536 blocks 895 -> 677 GB/s, 131 072 blocks 907 -> 657 GB/s end to end -- a chunk is requested only when the
     * walk is within 128 bytes of the end of what it has, and waits for it.  Nor do 64-byte chunks into 256-byte rings
     * (two request turns per round; measured 902 -> 818 and 931 -> 746 GB/s): more resident waves make this pass slower,
     * not faster, so it is not only a lone wave's instruction issue that paces it. */
    static const uint32_t ring_env = getenv("CRYO_LZ4_IDX_RING") ? (uint32_t)atoi(getenv("CRYO_LZ4_IDX_RING")) : 0u; /* tuning aid */
    const uint32_t ring = ring_env ? ring_env : 512u;
    const dim3 g((uint32_t)((n_blocks + kIdxLanes - 1) / kIdxLanes));
    if (ring == 256u) hipLaunchKernelGGL(k_lz4_index<256>, g, dim3(64), 0, s, d_src, d_src_off, d_src_size, n_blocks, tbl, cap, tbl_n);
    else hipLaunchKernelGGL(k_lz4_index<512>, g, dim3(64), 0, s, d_src, d_src_off, d_src_size, n_blocks, tbl, cap, tbl_n);
    return hipGetLastError();
}

hipError_t launch_lz4_decompress_indexed(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                                         const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                                         uint32_t block_size, uint64_t n_blocks, int32_t *d_status, void *d_workspace,
                                         size_t workspace_bytes)
{
    if (n_blocks == 0) return hipSuccess;
    const uint64_t grid = (n_blocks + 3) / 4;
    if (grid > 0x7fffffffull || workspace_bytes < lz4_index_workspace(n_blocks, block_size) || !d_workspace) return hipErrorInvalidValue;
    const uint32_t cap = lz4_index_cap(block_size);
    uint16_t *tbl = static_cast<uint16_t *>(d_workspace);
    uint32_t *tbl_n = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(d_workspace) + (((size_t)n_blocks * cap * 2u + 1024u + 15u) & ~(size_t)15u));
    if (hipError_t e = launch_lz4_index(s, d_src, d_src_off, d_src_size, n_blocks, tbl, cap, tbl_n); e != hipSuccess) return e;
    const dim3 g((uint32_t)grid), b(256);
    static const bool want_stats = getenv("CRYO_LZ4_STATS") != nullptr; /* debugging aid */
    if (want_stats) {
        unsigned long long *d_st = nullptr, h_st[32];
        if (hipMalloc((void **)&d_st, sizeof h_st) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d_st, 0, sizeof h_st, s);
        static const unsigned long long abl = getenv("CRYO_LZ4_ABLATE") ? strtoull(getenv("CRYO_LZ4_ABLATE"), nullptr, 0) : 0ull; /* timing experiments: wrong bytes */
        (void)hipMemcpyAsync(d_st + 7, &abl, sizeof abl, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL((k_lz4_dec_seq<4096, true>), g, b, 0, s, d_src, d_src_off, d_src_size, d_dst, dst_stride,
                           block_size, n_blocks, d_status, d_st, tbl, cap, tbl_n);
        (void)hipMemcpyAsync(h_st, d_st, sizeof h_st, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        (void)hipFree(d_st);
        fprintf(stderr, "[lz4 seq stats] batches %llu batch_seqs %llu general_seqs %llu match chunks %llu rounds %llu zero_batches %llu\n",
                h_st[0], h_st[1], h_st[2], h_st[3], h_st[4], h_st[5]);
        unsigned long long tot = 0;
        for (int k = 0; k < 8; k++) tot += h_st[8 + k];
        static const char *nm[8] = {"stage", "decode+validate", "flush", "requests", "lane runs+bitmap", "-", "match space", "general+other"};
        for (int k = 0; k < 8; k++) fprintf(stderr, "[lz4 seq cycles] %-16s %5.1f%%\n", nm[k], 100.0 * (double)h_st[8 + k] / (double)(tot ? tot : 1));
        fprintf(stderr, "[lz4 seq stops] not a candidate %llu, offset outside the window %llu, chain %llu, literal 255-run %llu, match 255-run %llu, "
                        "overlapping match %llu, offset too far %llu, end of input %llu, batch full (T) %llu, end of output %llu, far and long %llu\n",
                h_st[16], h_st[17], h_st[18], h_st[19], h_st[20], h_st[21], h_st[22], h_st[23], h_st[24], h_st[25], h_st[26]);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((k_lz4_dec_seq<4096, false>), g, b, 0, s, d_src, d_src_off, d_src_size, d_dst, dst_stride,
                       block_size, n_blocks, d_status, nullptr, tbl, cap, tbl_n);
    return hipGetLastError();
}

} // namespace cryo
