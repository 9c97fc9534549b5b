/*
 * lz4_dec.hip -- LZ4 block decode, one wavefront (64 lanes) per cryo block.
 *
 * Replaces LZ4_decompress_safe(compressed, out, compressed_size, CRYO_BLCKSZ)
 * (reference compression.c:84) for a batch of independent blocks.
 *
 * Accept/reject rules are those of the LZ4 block format as liblz4 1.9.3's
 * bounds-checked decoder applies them (csize = compressed size, B = block size):
 *   - a literal run reaching within 12 bytes of B or within 8 bytes of csize must be
 *     the final one: it has to end the input exactly and fit the output;
 *   - a literal-length extension byte at position q needs q + 15 < csize (terminator)
 *     and q + 16 < csize (a 255 that continues);
 *   - a match-length extension byte at position q needs q + 5 < csize;
 *   - the offset may not reach before the start of the output; offset 0 is not
 *     rejected by liblz4 1.9.3, it reproduces as zero bytes;
 *   - a match must end at least 5 bytes before B;
 * additionally a block must decode to exactly block_size bytes.
 *
 * Data movement of one wave (k_lz4_dec_ring):
 *
 *   HBM --16 B/lane, 1 KiB per wave-instruction--> s_in  (2 KiB input ring, LDS)
 *   s_in --one byte per lane--> `win`: a 64-byte look-ahead window in VGPRs
 *        lane 0 holds the token, the following lanes already hold the
 *        literals, the 2 offset bytes and the length-extension bytes, so one
 *        LDS read serves the whole parse of a typical sequence (v_readlane)
 *        and the literal copy needs no second read.
 *   win / s_ring --> s_ring  (output ring in LDS, the last R bytes produced):
 *        literals are written from the window registers; a match reads its
 *        source from the ring (LDS round trip instead of a global one).  LDS
 *        operations of one wave execute in order, so a match may read bytes
 *        written by the instruction before it.
 *   s_ring --16 B/lane--> HBM: every completed 1 KiB of output is flushed with
 *        one ds_read_b128 + one global_store_dwordx4 per lane.
 *
 * A match whose offset exceeds the ring reads its source back from the output
 * in HBM/L2 (already flushed, and vector memory operations of one wave are
 * executed in order).  Overlapping matches (offset < length, e.g. the zero gap
 * of a cryo block is a single ~110 KB match) are periodic: the first 64 bytes
 * use lane % offset, later pieces use a multiple of the offset >= 64.
 *
 * Two parse engines share this data movement:
 *
 *  BATCH (lz4_batch): the CU has ONE scalar unit (1 SALU instruction per cycle for
 *  all its waves) and a wave-uniform token parse saturates it (measured: 69 SALU per
 *  sequence, profiles/r01_v2_scalar_parse_pmc.json).  So the parse is done by the
 *  lanes instead: for a window of W compressed bytes every lane computes, for "its"
 *  byte positions, the distance d1 to the next token *if* a token started there (a
 *  256-entry token table in LDS does most of it); two doubling passes give d2, d4;
 *  a 16-hop chase over d4 plus a 3-step fill over d1 puts the start of sequence i
 *  into lane i.  Each lane then decodes one sequence's
 *  (literal length, offset, match length); a wave scan turns lengths into output
 *  positions; and the copy runs one output byte per lane, 64 bytes per step: a bitmap
 *  of sequence starts + mbcnt gives every byte its sequence, the byte's source is
 *  either the input ring (literal) or the output ring (match).  Chunks are produced
 *  in order, so only sources inside the current 64-byte chunk can be unready; those
 *  are resolved in extra rounds guarded by a ballot of finished lanes.
 *
 *  GENERAL (one sequence, wave-uniform parse): anything the batch does not take --
 *  length fields with 255-runs, literal runs > 58, overlapping or far matches, the
 *  first/last bytes of a block, malformed input.  It implements every accept/reject
 *  rule; the batch only ever accepts sequences that pass all of them.
 */
#include "lz_common.h"
#include "lz4_copy.h"
#include <cstdio>
#include <cstdlib>

#ifndef LZ4_WAVES_PER_SIMD
#define LZ4_WAVES_PER_SIMD 4 /* LDS admits 4 workgroups of 4 waves per CU: let the register allocator use what that leaves */
#endif

namespace cryo {

namespace {
} // namespace

/* ---- batch engine constants ---- */
constexpr uint32_t kWMax = 1024;  /* sequences of a batch start at window offsets < W <= kWMax */
constexpr uint32_t kD1N = kWMax + 384; /* d1 domain [0, W+256); d2 lookups reach < W+128+255 (a "bad" 255 is followed too) */
constexpr uint32_t kD2N = kWMax + 256; /* d2 domain [0, W+128); d4 lookups reach < W+255                         */
constexpr uint32_t kDBad = 255;   /* table value for "no simple token (run) starts here"                */
constexpr uint32_t kDMax = 63;    /* longest token-to-token distance the batch handles          */

/* Per-wave LDS regions.  They are separate __shared__ objects on purpose: the compiler can
 * then prove that e.g. a byte store into the output ring does not alias the parse tables,
 * and hoists the next group's LDS reads above it (otherwise every group costs a full LDS
 * round trip and the batch is latency bound). */
template <uint32_t R>
/* Every table pass is "all lanes store, then lanes read what OTHER lanes stored".  The hardware keeps a
 * wave's LDS operations in order; this keeps the compiler from moving the reads above the stores (seen
 * with the uniform-address reads of the chase: stale d4 entries, wrong sequence starts). */
#define LDS_TABLE_FENCE() asm volatile("" ::: "memory")

struct WaveLds {
    uint8_t *__restrict__ ring;            /* R        output ring                       */
    uint8_t *__restrict__ in;              /* kInRing  input ring                        */
    uint8_t *d1;                           /* kD1N  (phase 3 reuses it as `meta`)         */
    uint8_t *__restrict__ d2;              /* kD2N                                        */
    uint8_t *__restrict__ d4;              /* kWMax                                       */
    uint32_t *mmeta;                       /* 64: match meta of the copy engine (lz4_copy.h); lies over d1, dead by then */
    uint32_t *bm;                          /* kTMax/32 + 16: the copy engine's match-space bitmap and chunk bases */
    const uint8_t *lut;                    /* 256: token -> distance to the next token (see lz4_token_lut) */
};

/*
 * Decode up to 64 "simple" sequences starting at virtual position vp; NG*64 (<= kWMax) is the
 * span of compressed bytes searched for sequence starts.
 * Returns the number of sequences consumed (0 = none; caller runs the general path) and the
 * compressed bytes consumed in *used.
 *
 * Offsets: a match source is "near" (off < kNear, still in the LDS ring while the batch
 * writes up to kTMax new bytes) or "far" (off >= kNear: older than anything the batch
 * produces and already flushed, so it is read back from the output buffer in HBM/L2; all
 * far bytes of a batch are requested up front and cost one memory latency per batch).
 *
 * Every phase is written "all loads, then all arithmetic, then all stores" over fully
 * unrolled register arrays so that a phase costs ~one LDS round trip, not one per group.
 */
template <uint32_t R, uint32_t NG>
__device__ inline uint32_t lz4_batch(Wave<R> &w, const WaveLds<R> &L, uint32_t &vp, const uint32_t B,
                                     uint32_t *used, Stats &st)
{
    static_assert(R >= 4096, "ring must hold kTMax new bytes plus the near window");
    static_assert(NG * 64u <= kWMax, "window too large");
    constexpr uint32_t W = NG * 64u;
    /* groups of 64 in the d1 / d2 domains: d4 on W needs d2 on W + 126, which needs d1 on W + 128 + 63 */
    constexpr uint32_t G1 = NG + 3u, G2 = NG + 2u;
    const uint32_t lane = w.lane;
    const uint32_t vend = w.vend;
    *used = 0;
    if (vend < 32u || B < 32u) return 0;
    const uint32_t vsafe = vend - 16u;
    if (vp + 64u > vsafe || w.op + 64u > B) return 0;

    stamp(st, 7);
    /* stage the window (no refill happens while the batch runs) */
    while (w.in_hi < vend && w.in_hi < vp + W + 256u + 72u) w.refill();
    LDS_TABLE_FENCE(); /* the window below is read by other lanes than the ones that staged it */
    uint2 mir = make_uint2(0, 0);
    if (lane < 2u) mir = *reinterpret_cast<const uint2 *>(L.in + lane * 8u);
    else if (lane < 4u) mir = *reinterpret_cast<const uint2 *>(L.ring + (lane - 2u) * 8u);
    stamp(st, 0);

    /* ---- phase 1: d1 for every window offset, then d2, d4 ---- */
    uint32_t a1[G1], a2[G2];
    {
        uint32_t t[G1], e1[G1], tl[G1];
#pragma unroll
        for (uint32_t g = 0; g < G1; g++) {
            const uint32_t pos = vp + g * 64u + lane;
            t[g] = L.in[pos & kInMask];
            e1[g] = L.in[(pos + 1u) & kInMask];
        }
#pragma unroll
        for (uint32_t g = 0; g < G1; g++) tl[g] = L.lut[t[g]];
#pragma unroll
        for (uint32_t g = 0; g < G1; g++) {
            /* lut[token] = literal length + token + offset (+1 if the match length is extended); tokens whose
             * literal length is extended carry +64 and need the extension byte's value (one byte: larger -> bad) */
            const uint32_t b = tl[g];
            const uint32_t dx = b + e1[g] - 63u; /* (b - 64) + extension byte + its value */
            uint32_t d = b > 63u ? dx : b;
            if (d > kDMax) d = kDBad; /* "not batchable from here": saturates every sum it enters */
            a1[g] = d;
        }
#pragma unroll
        for (uint32_t g = 0; g < G1; g++) L.d1[g * 64u + lane] = (uint8_t)a1[g];
    }
    LDS_TABLE_FENCE();
    {
        uint32_t b[G2];
#pragma unroll
        for (uint32_t g = 0; g < G2; g++) b[g] = L.d1[g * 64u + lane + a1[g]]; /* a bad entry reads 255 further on: inside the array, and its sum saturates anyway */
#pragma unroll
        for (uint32_t g = 0; g < G2; g++) { const uint32_t v = a1[g] + b[g]; a2[g] = v < kDBad ? v : kDBad; } /* valid <= 126 */
#pragma unroll
        for (uint32_t g = 0; g < G2; g++) L.d2[g * 64u + lane] = (uint8_t)a2[g];
    }
    LDS_TABLE_FENCE();
    uint32_t a4[NG];
    {
        uint32_t b[NG];
#pragma unroll
        for (uint32_t g = 0; g < NG; g++) b[g] = L.d2[g * 64u + lane + a2[g]];
#pragma unroll
        for (uint32_t g = 0; g < NG; g++) { const uint32_t v = a2[g] + b[g]; a4[g] = v < kDBad ? v : kDBad; } /* valid <= 252 */
    }
    /* d4 stays in registers, four window positions per dword: position p = 64 g + lane sits in byte (g & 3) of
     * word g >> 2 of lane `lane` */
    constexpr uint32_t NW = (NG + 3u) / 4u;
    uint32_t pk[4] = {~0u, ~0u, ~0u, ~0u};
#pragma unroll
    for (uint32_t g = 0; g < NG; g++) pk[g >> 2] = (pk[g >> 2] & ~(255u << (8u * (g & 3u)))) | (a4[g] << (8u * (g & 3u)));
    stamp(st, 1);
    /* ---- chase: start of every 4th sequence into lanes 0,4,8,... ----
     * 16 serial hops over d4.  A hop used to be a wave-uniform LDS read: ~300 cycles each with the CU's waves
     * hammering the LDS, 4.8 k cycles per batch, a third of its time.  Now v_readlane from the registers above and a
     * few scalar instructions. */
    uint32_t sl = 0;   /* window offset of this lane's sequence */
    uint32_t ngrp = 0; /* groups of 4 sequences found */
    {
        uint32_t sv = 0, s0 = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16u; k++) {
            asm("v_writelane_b32 %0, %1, %2" : "+v"(sv) : "s"(s0), "i"(4u * k)); /* start of group k -> lane 4k */
            if (ngrp == k && s0 < W) {
                const uint32_t ln = s0 & 63u, wsel = s0 >> 8;
                uint32_t word = lane_get(pk[0], ln);
                if (NW > 1u && wsel == 1u) word = lane_get(pk[1], ln);
                if (NW > 2u && wsel == 2u) word = lane_get(pk[2], ln);
                if (NW > 3u && wsel == 3u) word = lane_get(pk[3], ln);
                const uint32_t dd = (word >> (((s0 >> 6) & 3u) * 8u)) & 255u;
                if (dd != kDBad) { s0 += dd; ngrp = k + 1u; }
            }
        }
        sl = (uint32_t)__builtin_amdgcn_mov_dpp((int)sv, 0x00, 0xf, 0xf, true); /* quad_perm [0,0,0,0]: lane 4k's value to its quad */
    }
    if (ngrp == 0u) return 0;
    const uint32_t ncand = ngrp * 4u;
    /* fill: lanes 4a+r walk r tokens from the group start */
#pragma unroll
    for (uint32_t hp = 0; hp < 3u; hp++) {
        const uint32_t dd = L.d1[sl];
        if ((lane & 3u) > hp && lane < ncand) sl += dd; /* lanes past the last usable group stay put */
    }

    stamp(st, 2);
    /* ---- phase 2: one sequence per lane ---- */
    const uint32_t pos = vp + sl;
    const uint32_t t = L.in[pos & kInMask];
    const uint32_t e1 = L.in[(pos + 1u) & kInMask];
    uint32_t ll = t >> 4;
    uint32_t k = 1u;
    if (ll == 15u) { ll += e1; k = 2u; }
    const uint32_t q = pos + k + ll; /* offset field */
    const uint32_t off = (uint32_t)L.in[q & kInMask] | ((uint32_t)L.in[(q + 1u) & kInMask] << 8);
    const uint32_t e2 = L.in[(q + 2u) & kInMask];
    uint32_t ml = (t & 15u) + 4u;
    uint32_t dlen = k + ll + 2u;
    const bool hasM = (t & 15u) == 15u;
    if (hasM) { ml += e2; dlen += 1u; }
    const bool cand = lane < ncand;
    const uint32_t outlen = cand ? ll + ml : 0u;
    const uint32_t oend = scan64_incl(outlen);
    const uint32_t ostart = oend - outlen;
    const uint32_t mabs = w.op + ostart + ll; /* absolute output position of the match */
    const bool isfar = cand && off >= R - kTMax; /* R - T >= T + 1023: in the ring for the whole batch, or flushed before it */
    const bool ok = cand && !(hasM && e2 == 255u) && (off >= ml || (off != 0u && ml <= 64u)) /* short self-overlap: a dependent match, lz4_copy.h */ && off <= mabs &&
                    pos + dlen <= vsafe && oend <= kTMax && w.op + oend + 16u <= B && !(isfar && ml > 32u);
    const unsigned long long badmask = wave_ballot(!ok);
    const uint32_t nseq = badmask ? ctz64(badmask) : 64u;
    /* the output ring's first 16 bytes are mirrored behind it, like the input ring's (16-byte reads of the copy
     * engine that start in a ring's last 15 bytes) */
    if (lane < 2u) *reinterpret_cast<uint2 *>(L.in + kInRing + lane * 8u) = mir;
    else if (lane < 4u) *reinterpret_cast<uint2 *>(L.ring + R + (lane - 2u) * 8u) = mir;
    w.flush();   /* what earlier batches produced: far sources are read back from it */
    if (nseq == 0u) return 0;
    const uint32_t T = lane_get(oend, nseq - 1u);
    *used = lane_get(sl + dlen, nseq - 1u);
    /* far matches (source may leave the ring before the batch is done; flushed): two 16-byte requests per lane */
    uint4 xfa = make_uint4(0, 0, 0, 0), xfb = xfa;
    if (lane < nseq && isfar) {
        const uint8_t *g = w.dst + (mabs - off);
        __builtin_memcpy(&xfa, g, 16);
        __builtin_memcpy(&xfb, g + 16, 16);
    }
    stamp(st, 3);
    const CopyLds<R, kTMax> SL = {L.ring, L.in, L.mmeta, L.bm};
    seq_copy<R, kTMax>(w, SL, nseq, ostart, ll, ml, off, pos + k, T, isfar, xfa, xfb, st);
    vp += *used;
    stamp(st, 0);
    return nseq;
}


template <uint32_t R, bool STATS>
__global__ void __launch_bounds__(256, LZ4_WAVES_PER_SIMD) /* VGPR cap matching what the LDS budget admits */
k_lz4_dec_ring(const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
               const uint32_t *__restrict__ src_size, uint8_t *dst_base, uint64_t dst_stride, uint32_t B,
               uint64_t n_blocks, int32_t *__restrict__ status, unsigned long long *stats, uint32_t only_heavy,
               const uint32_t *__restrict__ decoded)
{
    Stats st = {};
    st.on = STATS;
    if (STATS) { st.ablate = (uint32_t)stats[7]; st.t0 = __builtin_amdgcn_s_memtime(); }
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[4][R + 16];      /* + the 16-byte tail of the copy engine (lz4_copy.h) */
    __shared__ __attribute__((aligned(16))) uint8_t s_in[4][kInRing + 16];
    __shared__ __attribute__((aligned(16))) uint8_t s_d1[4][kD1N]; /* also holds meta[64] in phase 3 */
    __shared__ uint8_t s_d2[4][kD2N];
    __shared__ uint8_t s_d4[4][kWMax];
    __shared__ __attribute__((aligned(8))) uint32_t s_bm[4][kTMax / 32 + 16]; /* bitmap + per-chunk bases */
    __shared__ __attribute__((aligned(4))) uint8_t s_lut[256]; /* every wave writes the same values before it reads them */

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = uni(threadIdx.x >> 6);
    const uint64_t blk = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wid;
    if (blk >= n_blocks) return;

    /* wave-uniform values are forced into SGPRs so the parse runs on the scalar unit */
    const uint8_t *base = src_base + uni64(src_off[blk]);
    const uint32_t csize = uni(src_size[blk]);
    if (only_heavy != 0u && !lz4_literal_heavy(csize, B)) return; /* the indexed decoder has this block (kernels.h) */
    if (decoded != nullptr && uni(decoded[blk]) != 0u) return;        /* by the few-blocks path (lz4_lat.hip) */

    Wave<R> w;
    const WaveLds<R> L = {s_ring[wid], s_in[wid], s_d1[wid], s_d2[wid], s_d4[wid],
                          reinterpret_cast<uint32_t *>(s_d1[wid]), s_bm[wid], s_lut};
    {
        uint32_t v = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            const uint32_t tok = lane * 4u + k, hi = tok >> 4;
            const uint32_t d = hi + 3u + ((tok & 15u) == 15u ? 1u : 0u);
            v |= (hi == 15u ? d + 64u : d) << (8u * k);
        }
        reinterpret_cast<uint32_t *>(s_lut)[lane] = v;
        LDS_TABLE_FENCE();
    }
    w.ring = L.ring;
    w.in = L.in;
    w.lane = lane;
    w.delta = (uint32_t)(reinterpret_cast<uintptr_t>(base) & 15u);
    w.abase = base - w.delta;
    w.vend = w.delta + csize;
    w.in_hi = 0;
    w.dst = dst_base + uni64(blk * dst_stride);
    w.dst_aligned = (reinterpret_cast<uintptr_t>(w.dst) & 15u) == 0;
    w.op = 0;
    w.flushed = 0;

    /* all stream positions below are "virtual": vp = delta + offset in the compressed block */
    const uint32_t vend = w.vend;
    uint32_t vp = w.delta;
    bool bad = (csize == 0);
    bool done = bad;
    uint32_t skip = 0;
    uint32_t wmode = 1u; /* parse window: 0 = 128 B, 1 = 512 B, 2 = 1 KiB */

    if (!bad) {
        w.prefetch();
        w.refill();
        if (w.in_hi < vend) w.refill();
    }
    uint32_t win = bad ? 0u : w.window(vp);

    while (!done) {
        /* =====================  BATCH PATH  =====================
         * The window W follows the compressed bytes a batch actually consumed (long matches
         * make a batch hit kTMax output bytes after few tokens). */
        if (skip == 0u) {
            uint32_t n, used;
            do {
                n = wmode == 0u ? lz4_batch<R, 2>(w, L, vp, B, &used, st)
                  : (wmode == 1u ? lz4_batch<R, 8>(w, L, vp, B, &used, st)
                  : (wmode == 2u ? lz4_batch<R, 12>(w, L, vp, B, &used, st) : lz4_batch<R, 16>(w, L, vp, B, &used, st)));
                if (n == 0u) st.zero_batches++;
                /* the window follows the data: long matches fill kTMax output bytes after few tokens
                 * (128 bytes of input suffice), literal-heavy blocks need up to 1 KiB of input for 64 tokens;
                 * table work is proportional to the window, so it is kept just above what a batch consumes */
                const uint32_t wbytes = wmode == 0u ? 128u : (wmode == 1u ? 512u : (wmode == 2u ? 768u : 1024u));
                if (n != 0u && n < 64u && used < 96u) wmode = 0u;
                else if (n != 0u && n < 40u && used + 64u > wbytes && wmode < 3u) wmode++;
                else if (wmode == 3u && used < 640u) wmode = 2u;
                else if (wmode == 2u && used < 400u) wmode = 1u;
                else if (wmode == 0u && used >= 96u) wmode = 1u;
            } while (n >= 24u || (n >= 4u && wmode == 0u));
            if (n < 4u) skip = 8u; /* poor yield: stay on the general path for a while */
        } else {
            skip--;
        }

        /* =====================  GENERAL PATH: one sequence  ===================== */
        st.general_seqs++;
        w.flush();
        w.need(vp);
        win = w.window(vp);

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
                if (wbase + 63u + 16u >= vend) { bad = true; break; }
                ll += (64u - k) * 255u;
                wbase += 64u;
                w.need(wbase);
                win = w.window(wbase);
                k = 0;
            }
            if (bad) break;
            /* terminating byte at position pe must satisfy pe + 15 < csize */
            if (wbase + (k - 1u) + 15u >= vend) { bad = true; break; }
            vp = wbase; /* lane k of the current window is the byte after the field */
        }
        const uint32_t lp = vp + k; /* virtual position of the first literal */

        /* ---- literal run ---- */
        const bool last = (w.op + ll + 12u > B) || (lp + ll + 8u > vend);
        if (last && (lp + ll != vend || w.op + ll > B)) { bad = true; break; }
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
            if (last) { done = true; break; }
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
                if (wbase + 63u + 5u >= vend) { bad = true; break; }
                ml += (64u - k) * 255u;
                wbase += 64u;
                w.need(wbase);
                win = w.window(wbase);
                k = 0;
            }
            if (bad) break;
            if (wbase + (k - 1u) + 5u >= vend) { bad = true; break; }
            vp = wbase;
        }
        ml += 4u;
        if (off > w.op || w.op + ml + 5u > B) { bad = true; break; }

        /* ---- next window: issue its LDS read before the match copy so the two overlap ---- */
        vp += k;
        w.need(vp);
        win = w.window(vp);

        /* ---- match copy ---- */
        {
            uint32_t rem = ml;
            if (off == 0u) {
                /* liblz4 1.9.3 does not reject offset 0: it reproduces as zero bytes */
                while (rem) {
                    w.flush();
                    const uint32_t n = rem < 64u ? rem : 64u;
                    if (lane < n) w.ring[(w.op + lane) & (R - 1)] = 0;
                    w.op += n;
                    rem -= n;
                }
            } else if (off <= R - 128u) {
                /* source inside the LDS ring */
                uint32_t eff = off; /* distance used by pieces after the first */
                uint32_t sidx;      /* ring index this lane reads for the first piece */
                if (off < 64u && ml > off) {
                    sidx = w.op - off + lane_mod(lane, off);
                    /* smallest multiple of the period >= 64: out[i] = out[i - eff] holds for every
                     * i >= op + 64 only while eff <= 64 + off */
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
                    if (pow2_period && rem >= 2u * R && wave_stream_pattern(w, rem, false)) continue; /* long run: lz_common.h */
                    uint32_t n = rem < 64u ? rem : 64u;
                    if (pow2_period && rem >= 3u * R) { const uint32_t to = kChunk - (w.op & (kChunk - 1u)); n = n < to ? n : to; }
                    const uint8_t x = w.ring[(w.op - eff + lane) & (R - 1)];
                    if (lane < n) w.ring[(w.op + lane) & (R - 1)] = x;
                    w.op += n;
                    rem -= n;
                }
            } else {
                /* far match: source already flushed to the output buffer (off > R-128 >= 1088+) */
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

/* Which path a batch takes.  The indexed decoder (lz4_index.hip + lz4_dec2.hip) does half the work per sequence of
 * the in-wave parse below, but its index pass is a serial walk per walker: with one walker per block it takes as long
 * for 4 096 blocks as for 65 536 (and 8 x longer for 1 MiB blocks), so round 2 used it from 24 576 blocks on only.  With
 * several walkers per block the pass shrinks with the batch: the walkers are chosen so that the batch fills the chip
 * once (65 536 lanes = one wave per SIMD) with segments of at least 4 KiB of block.  Small blocks in small batches stay
 * on the in-wave parse (a 4 KiB block is a few batches: the second launch costs more than the parse). */
uint32_t lz4_decode_plan(uint64_t n_blocks, uint32_t block_size, const Lz4DecodeOpts &opts)
{
    if (opts.path == 1) return 0;
    if (opts.path == 0 && block_size < 16384u && n_blocks < 24576u) return 0;
    if (opts.walkers > 0) return (uint32_t)opts.walkers;
    uint32_t S = 1;
    while (S < 64u && n_blocks * (2u * S) <= kLz4IndexResidentLanes && block_size / (2u * S) >= 4096u) S *= 2u;
    return S;
}
/* the few-blocks path (lz4_lat.hip): asked for, or automatic for what it is made for */
static bool lz4_use_latency(uint64_t n_blocks, uint32_t block_size, const Lz4DecodeOpts &opts)
{
    if (!lz4_latency_eligible(n_blocks, block_size)) return false;
    return opts.path == 3 || (opts.path == 0 && opts.walkers == 0);
}

size_t lz4_decompress_workspace(uint64_t n_blocks, uint32_t block_size, const Lz4DecodeOpts &opts)
{
    if (lz4_use_latency(n_blocks, block_size, opts)) return lz4_latency_workspace(n_blocks, block_size);
    const uint32_t S = lz4_decode_plan(n_blocks, block_size, opts);
    return S ? lz4_index_layout(n_blocks, block_size, S).bytes : 0;
}

hipError_t launch_lz4_decompress(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                                 const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                                 uint32_t block_size, uint64_t n_blocks, int32_t *d_status, void *d_workspace,
                                 size_t workspace_bytes, const Lz4DecodeOpts &opts)
{
    if (n_blocks == 0) return hipSuccess;
    const uint64_t grid = (n_blocks + 3) / 4;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    if (lz4_use_latency(n_blocks, block_size, opts))
        return launch_lz4_decompress_latency(s, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size, n_blocks, d_status,
                                             d_workspace, workspace_bytes);
    /* sequence index pass + the decoder built for it (lz4_dec2.hip) */
    const uint32_t S = lz4_decode_plan(n_blocks, block_size, opts);
    if (S != 0) {
        if (d_workspace == nullptr) return hipErrorInvalidValue;
        return launch_lz4_decompress_indexed(s, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size, n_blocks,
                                             d_status, d_workspace, workspace_bytes, S, opts.waves, &opts);
    }
#ifdef CRYO_DEBUG
    static const bool want_stats = cryo_tuning_env("CRYO_LZ4_STATS") != nullptr; /* debugging aid */
    if (want_stats) {
        const dim3 g((uint32_t)grid), b(256);
        unsigned long long *d_st = nullptr, h_st[16];
        static const unsigned long long abl = cryo_tuning_env("CRYO_LZ4_ABLATE") ? strtoull(cryo_tuning_env("CRYO_LZ4_ABLATE"), nullptr, 0) : 0ull;
        if (hipMalloc((void **)&d_st, sizeof h_st) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d_st, 0, sizeof h_st, s);
        (void)hipMemcpyAsync(d_st + 7, &abl, sizeof abl, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL((k_lz4_dec_ring<4096, true>), g, b, 0, s, d_src, d_src_off, d_src_size, d_dst, dst_stride,
                           block_size, n_blocks, d_status, d_st, 0u, nullptr);
        (void)hipMemcpyAsync(h_st, d_st, sizeof h_st, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        (void)hipFree(d_st);
        if (!cryo_tuning_env("CRYO_LZ4_QUIET")) fprintf(stderr, "[lz4 stats] batches %llu batch_seqs %llu general_seqs %llu chunks %llu rounds %llu zero_batches %llu\n",
                h_st[0], h_st[1], h_st[2], h_st[3], h_st[4], h_st[5]);
        if (!cryo_tuning_env("CRYO_LZ4_QUIET")) {
            unsigned long long tot = 0;
            for (int k = 0; k < 8; k++) tot += h_st[8 + k];
            static const char *nm[8] = {"stage+flush", "phase1 tables", "chase+fill", "phase2 meta", "phase3 bitmap", "passA", "passB", "general+other"};
            for (int k = 0; k < 8; k++) fprintf(stderr, "[lz4 cycles] %-14s %5.1f%%\n", nm[k], 100.0 * (double)h_st[8 + k] / (double)(tot ? tot : 1));
        }
        return hipGetLastError();
    }
#endif
    return launch_lz4_dec_ring(s, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size, n_blocks, d_status, false);
}

hipError_t launch_lz4_dec_ring(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off, const uint32_t *d_src_size,
                               uint8_t *d_dst, uint64_t dst_stride, uint32_t block_size, uint64_t n_blocks, int32_t *d_status,
                               bool only_literal_heavy, const uint32_t *d_done)
{
    const uint64_t grid = (n_blocks + 3) / 4;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_lz4_dec_ring<4096, false>), dim3((uint32_t)grid), dim3(256), 0, s, d_src, d_src_off, d_src_size, d_dst,
                       dst_stride, block_size, n_blocks, d_status, nullptr, only_literal_heavy ? 1u : 0u, d_done);
    return hipGetLastError();
}

} // namespace cryo
