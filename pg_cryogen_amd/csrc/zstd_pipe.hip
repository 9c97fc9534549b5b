/*
 * zstd_pipe.hip -- Zstandard frame decode for large batches: four kernels, each shaped after the
 * parallelism its stage actually has.
 *
 * Replaces ZSTD_decompress(out, CRYO_BLCKSZ, compressed, compressed_size) (reference
 * compression.c:116) for a batch of independent cryo blocks, like the fused decoder in
 * zstd_dec.hip, and gives bit-identical results and statuses.  The fused decoder spends ~90% of its
 * time in two strictly serial bitstream loops (Huffman literals on 1-4 lanes, FSE sequences on one
 * lane-equivalent) while the other lanes of the wave idle.  Across a batch those loops are
 * embarrassingly parallel, so here they run one LANE per stream:
 *
 *   K1  k_zplan   wave per frame   frame/block/section headers -> descriptors; Huffman and FSE decoding tables built by
 *                                  the whole wave, straight into the workspace
 *   K2  k_zhufw   wave per block   Huffman literals, 16 walkers per stream from guessed bit positions (a prefix code
 *                                  resynchronises), two symbols per lookup; k_zhuf (lane per stream, 16 blocks' tables
 *                                  in LDS) takes what the walkers hand back
 *   K3  k_zchain4 quad per block   the serial part of the FSE sequence stream only (the three states on three lanes of a
 *                                  quad): 16 blocks' tables (16-bit entries) in LDS, 8 bytes per sequence out (bit
 *                                  position, the three states)
 *   K3' k_zmat    wave per frame   values (base + extra bits) and repeat offsets of 512 sequences per round: the offset
 *                                  history by a scan over "history transforms"
 *   K4  k_zexec   wave per frame   sequence execution with the shared LZ copy engine (lz_common.h): 8-byte records
 *                                  loaded 64 at a time, literals stream through the LDS input ring, output ring in
 *                                  LDS, 1 KiB coalesced flushes
 *
 * Anything K1 does not recognise as "one well-formed frame of at most nbmax blocks" (concatenated or
 * skippable frames, malformed headers, pool exhaustion) is put on an irregular list and decoded by the
 * fused kernel afterwards, so coverage and error behaviour are exactly the fused decoder's.
 *
 * The batch is processed in tiles of F <= 14848 frames (one full round of K3 on 256 CUs), two tiles in flight, so the
 * workspace stays bounded.
 */
#include "zstd_common.h"
#include "lz4_copy.h"
#include "kernels.h"
#include "lat_copy.h"
#include <cstdio>
#include <cstdlib>

namespace cryo {

namespace {

constexpr uint32_t kSeqTblWords = 1280; /* LL 512 | ML 512 | OF 256 */
constexpr uint32_t kHufTblWords = 4096; /* u16 entries per block slot */
constexpr uint32_t kPredefSlot = 255;
constexpr uint32_t F_BAD = 1u, F_IRREG = 2u, F_CK = 4u, F_FCS = 8u;

struct ZBlk { /* one zstd block of a frame; 128 bytes */
    uint32_t type;     /* 0 raw, 1 RLE, 2 compressed */
    uint32_t src_off;  /* block content, offset in the frame's input */
    uint32_t bsize;
    uint32_t lit_mode; /* 0 raw bytes in the input, 1 RLE, 2 Huffman -> literal pool */
    uint32_t lit_src;  /* mode 0: input offset; 1: the byte; 2: offset in the frame's literal pool */
    uint32_t regen;
    uint32_t nstreams;
    uint32_t hs_off[4], hs_len[4]; /* Huffman streams (input offsets) */
    uint32_t huf_slot, huf_log;
    uint32_t nseq;
    uint32_t sq_off, sq_len; /* sequence bitstream */
    uint32_t slots;          /* table slots: ll | of << 8 | ml << 16 */
    uint32_t logs;           /* table logs, same packing */
    uint32_t seq_base;       /* first record in the sequence pool */
    uint32_t segd;           /* Huffman literals: 1 = they lie in the walkers' pieces of scratch (ZPipe::hsegs), k_zexec reads them in place */
    uint32_t pad[8];
};
static_assert(sizeof(ZBlk) == 128, "descriptor size");

struct ZFrame {
    uint32_t nblk, flags, fcs_lo, fcs_hi, ck_off, pad[3];
};

struct ZPipe {
    const uint8_t *src_base;
    const uint64_t *src_off;
    const uint32_t *src_size;
    uint8_t *dst_base;
    uint64_t dst_stride;
    uint32_t B;
    uint64_t first; /* first block of the tile */
    uint32_t F, nbmax, litcap, seqcap;
    int32_t *status;
    ZFrame *frames;
    ZBlk *blks;
    uint16_t *huf;
    uint32_t *seqt, *predef;
    uint8_t *lits;
    uint2 *seqs;  /* offset (29 bits, larger ones as 2^29 - 1) | literal length << 29 (17 bits), match length << 14 (18 bits) */
    uint2 *chain; /* k_zchain's records: unread bits before the sequence (20) | LL state << 20, OF state | ML state << 8 */
    uint32_t *counters; /* [0] sequence pool cursor, [1] Huffman items, [2] irregular frames, [3] sequence items,
                           [61] blocks k_zhufw hands back */
    uint32_t *hitems;
    uint32_t *hitems2; /* blocks k_zhufw hands back to k_zhuf (counters[61]) */
    uint8_t *htmp;     /* k_zhufw's scratch: the walkers' symbols before they are moved to the literal pool */
    uint64_t htmp_stride;
    uint4 *hsegs;      /* per block 64 entries, one per walker: scratch offset of its true symbols, their count, pool offset (k_zmove / k_zexec) */
    uint32_t in_place; /* 1: k_zexec reads the walkers' pieces where they lie (no k_zmove) unless a block has a piece shorter than 8 bytes */
    uint32_t *mitems;  /* blocks whose symbols k_zmove moves (counters[60]) */
    uint32_t *irregular;
    uint32_t *sitems; /* blocks with sequences (f * nbmax + k) */
    const uint32_t *done; /* few frames per call: frames the byte-parallel execution has decoded (k_zexec skips them); else nullptr */
    uint32_t hufw_seglog; /* k_zhufw: a stream of T bits gets T >> this walkers (1 .. 16) */
    uint32_t hufw_min; /* Huffman blocks of fewer literals than this go to k_zhuf (lane per stream) instead of k_zhufw's walkers */
};

struct PlanLds {
    uint32_t ll[64], ml[64], of[64]; /* an RLE table (one entry) or a predefined one (log <= 6); described tables are built straight into the workspace */
    int16_t norm3[3][64];   /* per table kind (LL, OF, ML): normalized counts, next-state counters, cell symbols */
    uint16_t nxt3[3][64];
    uint8_t cell3[3][512];
    uint8_t hw[256]; /* staged: first bytes of a block (literals header, Huffman description, jump table) */
    uint8_t sw[256]; /* staged: first bytes of its sequences section (count, modes, table descriptions) */
    int16_t norm[256];
    uint16_t nxt[256];
    uint32_t wdt[64];
    uint8_t wts[256];
    uint8_t cell[512];
};

struct PlanState {
    bool huf_valid, fse_valid;
    uint32_t huf_slot, huf_log;
    uint32_t slot[3], log[3]; /* LL, OF, ML */
};

__device__ inline void copy_words(uint32_t *dst, const uint32_t *src, uint32_t n, uint32_t lane)
{
    for (uint32_t i = lane; i < n; i += 64u) dst[i] = src[i];
}

/* sequence decoding table -> workspace, with the symbol's extra-bit count folded in at bits 20..24 (K3 needs
 * the bit counts, not the base values, to find the next state: one LDS round trip less on its critical path) */
__device__ inline void copy_seq_table(uint32_t *dst, const uint32_t *src, uint32_t n, int kind, uint32_t lane)
{
    for (uint32_t i = lane; i < n; i += 64u) {
        const uint32_t e = src[i], sym = e >> 14;
        const uint32_t xb = kind == 0 ? kLLBits[sym] : (kind == 1 ? sym : kMLBits[sym]);
        dst[i] = e | (xb << 20);
    }
}

/* 256 bytes from global memory into LDS in one coalesced load: the header parsers read single bytes, each
 * of which would otherwise be a dependent global load (~150 of them per block) */
__device__ inline void stage256(uint8_t *lds, const uint8_t *g, uint32_t avail, uint32_t lane)
{
    const uint32_t o = lane * 4u;
    uint32_t v = 0;
    if (o + 4u <= avail) __builtin_memcpy(&v, g + o, 4);
    else for (uint32_t k = 0; k < 4u; k++) if (o + k < avail) v |= (uint32_t)g[o + k] << (8u * k);
    reinterpret_cast<uint32_t *>(lds)[lane] = v;
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

/* One FSE decoding table, built by the whole wave (libzstd's ZSTD_buildFSETable order: low-probability symbols from the
 * top, then symbol after symbol at multiples of `step`; a cell's next-state number = the symbol's count + the cell's rank
 * among the symbol's cells).  One lane per symbol for the counts, per cell for everything else; the entry goes to the
 * workspace with the symbol's extra-bit count at bits 20..24 (copy_seq_table).  cell: 512 bytes, run / cum: 64 x u16. */
__device__ inline bool fse_build_wave(uint8_t *cell, uint16_t *run, uint16_t *cum, const int16_t *norm, const int max_sym, const int log,
                                      uint32_t *out, const int kind, const uint32_t lane)
{
    const uint32_t size = 1u << log, mask = size - 1u, step = (size >> 1) + (size >> 3) + 3u;
    const int c = (int)lane <= max_sym ? (int)norm[lane] : 0;
    const bool low = c == -1;
    const uint32_t pc = c > 0 ? (uint32_t)c : 0u;
    const unsigned long long lm = wave_ballot(low);
    const uint32_t nlow = (uint32_t)__builtin_popcountll(lm);
    const uint32_t incl = scan64_incl(pc);
    const uint32_t total = lane_get(incl, 63);
    if (total + nlow != size) return false;
    const uint32_t high = size - 1u - nlow;
    if (low) cell[size - 1u - __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u))] = (uint8_t)lane;
    cum[lane] = (uint16_t)(incl - pc);
    run[lane] = (uint16_t)(low ? 1u : pc);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    uint32_t vbase = 0;
    for (uint32_t j0 = 0; j0 < size; j0 += 64u) {
        const uint32_t j = j0 + lane;
        const uint32_t p = (j * step) & mask;
        const bool valid = j < size && p <= high;
        const unsigned long long m = wave_ballot(valid);
        const uint32_t i = vbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        vbase += (uint32_t)__builtin_popcountll(m);
        uint32_t lo = 0; /* the last symbol whose first rank is <= i: the i-th placed cell is its */
#pragma unroll
        for (uint32_t bstep = 32u; bstep >= 1u; bstep >>= 1) lo += cum[lo + bstep] <= i ? bstep : 0u;
        if (valid) cell[p] = (uint8_t)lo;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    for (uint32_t p0 = 0; p0 < size; p0 += 64u) {
        const uint32_t p = p0 + lane;
        const bool on = p < size;
        const uint32_t sym = on ? cell[p] : 0xFFFFu;
        unsigned long long rem = wave_ballot(on);
        uint32_t r = 0, cnt = 1;
        while (rem) {
            const uint32_t s0 = lane_get(sym, (uint32_t)__builtin_ctzll(rem));
            const unsigned long long m = wave_ballot(sym == s0);
            if (sym == s0) {
                r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                cnt = (uint32_t)__builtin_popcountll(m);
            }
            rem &= ~m;
        }
        const uint32_t ns = (on ? run[sym] : 1u) + r;
        asm volatile("" ::: "memory");
        if (on && r + 1u == cnt) run[sym] = (uint16_t)(ns + 1u);
        asm volatile("" ::: "memory");
        const uint32_t nb = (uint32_t)log - hb32(ns);
        const uint32_t xb = kind == 0 ? kLLBits[sym < 36u ? sym : 0u] : (kind == 1 ? sym : kMLBits[sym < 53u ? sym : 0u]);
        if (on) out[p] = ((ns << nb) - size) | (nb << 10) | (sym << 14) | (xb << 20);
    }
    __builtin_amdgcn_wave_barrier();
    return true;
}

#ifdef CRYO_HW_PROF
#define PL_STAMP(k) do { const uint64_t tn_ = __builtin_amdgcn_s_memtime(); if (lane == 0u) atomicAdd(&P.counters[k], (uint32_t)((tn_ - pl_t) >> 6)); pl_t = tn_; } while (0)
#define PL_BEGIN() uint64_t pl_t = __builtin_amdgcn_s_memtime()
#else
#define PL_STAMP(k) do { } while (0)
#define PL_BEGIN() do { } while (0)
#endif
/* K1: one compressed block's section headers -> descriptor + tables.  false = not plannable. */
__device__ bool plan_block(PlanLds &L, const ZPipe &P, PlanState &ps, const uint8_t *src, const uint8_t *hs /* staged src[0..253) */,
                           uint32_t n, uint32_t boff,
                           uint32_t f, uint32_t k, uint32_t &lit_cursor, ZBlk &d, uint32_t lane)
{
    if (n < 3u) return false;
    PL_BEGIN();
    const uint32_t b0 = uni(hs[0]);
    const uint32_t type = b0 & 3u, fmt = (b0 >> 2) & 3u;
    uint32_t regen, used;
    d.nstreams = 0;
    d.huf_slot = 0; d.huf_log = 0;
    if (type < 2u) {
        uint32_t hdr;
        if (fmt == 1u) { hdr = 2; regen = (b0 >> 4) | (uni(hs[1]) << 4); }
        else if (fmt == 3u) { hdr = 3; regen = (b0 >> 4) | (uni(hs[1]) << 4) | (uni(hs[2]) << 12); }
        else { hdr = 1; regen = b0 >> 3; }
        if (type == 0u) {
            if (hdr + regen > n || regen > kZBlockMax) return false;
            d.lit_mode = 0; d.lit_src = boff + hdr; used = hdr + regen;
        } else {
            if ((fmt == 3u && n < 4u) || regen > kZBlockMax || hdr + 1u > n) return false;
            d.lit_mode = 1; d.lit_src = uni(hs[hdr]); used = hdr + 1u;
        }
    } else {
        if (n < 5u) return false;
        const uint32_t h = b0 | (uni(hs[1]) << 8) | (uni(hs[2]) << 16) | (uni(hs[3]) << 24);
        uint32_t hdr, csize;
        bool single = false;
        if (fmt < 2u) { single = (fmt == 0u); hdr = 3; regen = (h >> 4) & 0x3FFu; csize = (h >> 14) & 0x3FFu; }
        else if (fmt == 2u) { hdr = 4; regen = (h >> 4) & 0x3FFFu; csize = h >> 18; }
        else { hdr = 5; regen = (h >> 4) & 0x3FFFFu; csize = (h >> 22) + (uni(hs[4]) << 10); }
        if (regen > kZBlockMax || csize + hdr > n) return false;
        const uint8_t *p = src + hdr;
        uint32_t left = csize;
        if (type == 3u) { if (!ps.huf_valid) return false; }
        else {
            int hlog = 0;
            /* the decoding table is filled straight into the workspace (8 KiB less LDS: twice the workgroups per CU) */
            /* the weights' bit reader reads the staged copy too (the description is at most 1 + 127 bytes behind a header of at
             * most 5: inside the window).  It used to read the stream in global memory, a trip per eight bytes; that the phase
             * took the same time either way (profiles/r05_zplan_lds_reader.txt) says k_zplan is bound by the ~14 K instructions
             * of a frame's serial table work across its many resident waves, not by any one wave's trips */
            const int t = huf_read_table(L, P.huf + ((uint64_t)f * P.nbmax + k) * kHufTblWords, hs + hdr, hs + hdr, left, &hlog, lane);
            if (t < 0) return false;
            ps.huf_valid = true;
            ps.huf_slot = k;
            ps.huf_log = (uint32_t)hlog;
            p += t; left -= (uint32_t)t;
        }
        d.huf_slot = ps.huf_slot; d.huf_log = ps.huf_log;
        const uint32_t pofs = boff + (uint32_t)(p - src);
        if (single) {
            d.nstreams = 1;
            d.hs_off[0] = pofs; d.hs_len[0] = left;
            d.hs_off[1] = d.hs_off[2] = d.hs_off[3] = pofs;
            d.hs_len[1] = d.hs_len[2] = d.hs_len[3] = 0;
        } else {
            if (left < 10u) return false;
            const uint8_t *jt = hs + (uint32_t)(p - src); /* at most 5 + 129 bytes in: inside the staged window */
            const uint32_t l1 = uni((uint32_t)jt[0] | ((uint32_t)jt[1] << 8));
            const uint32_t l2 = uni((uint32_t)jt[2] | ((uint32_t)jt[3] << 8));
            const uint32_t l3 = uni((uint32_t)jt[4] | ((uint32_t)jt[5] << 8));
            if (6u + l1 + l2 + l3 > left) return false;
            const uint32_t seg = (regen + 3u) / 4u;
            if (3u * seg > regen) return false;
            d.nstreams = 4;
            d.hs_off[0] = pofs + 6u; d.hs_len[0] = l1;
            d.hs_off[1] = pofs + 6u + l1; d.hs_len[1] = l2;
            d.hs_off[2] = pofs + 6u + l1 + l2; d.hs_len[2] = l3;
            d.hs_off[3] = pofs + 6u + l1 + l2 + l3; d.hs_len[3] = left - (6u + l1 + l2 + l3);
        }
        if (lit_cursor + regen > P.litcap) return false; /* the frame would decode to more than B bytes */
        d.lit_mode = 2; d.lit_src = lit_cursor;
        lit_cursor += (regen + 31u) & ~15u; /* 16 .. 31 spare bytes: k_zhufw finishes a block's literals with a whole 16-byte store */
        used = hdr + csize;
    }
    d.regen = regen;

    PL_STAMP(17);
    /* sequences section header */
    uint32_t left = n - used;
    if (left < 1u) return false;
    stage256(L.sw, src + used, left, lane);
    const uint8_t *ip = L.sw; /* parsed from the staged copy: count, modes and the three descriptions fit in 256 bytes */
    uint32_t nseq = uni(ip[0]);
    ip++; left--;
    d.sq_off = 0; d.sq_len = 0; d.slots = 0; d.logs = 0;
    if (nseq == 0u) {
        if (left != 0u) return false;
    } else {
        if (nseq > 0x7Fu) {
            if (nseq == 0xFFu) { if (left < 2u) return false; nseq = uni(ip[0]) + (uni(ip[1]) << 8) + 0x7F00u; ip += 2; left -= 2u; }
            else { if (left < 1u) return false; nseq = ((nseq - 0x80u) << 8) + uni(ip[0]); ip++; left--; }
        }
        if (left < 1u) return false;
        const uint32_t modes = uni(ip[0]);
        ip++; left--;
        uint32_t *gt = P.seqt + ((uint64_t)f * P.nbmax + k) * kSeqTblWords;
        /* The three table descriptions are parsed in stream order (wave-uniform, cheap); the tables they
         * describe are then built by lanes 0, 1, 2 at the same time -- same code, private scratch -- instead
         * of one after the other on the whole wave. */
        int bmode[3], bms[3], blg[3];
#pragma unroll
        for (int kind = 0; kind < 3; kind++) { /* LL, OF, ML in stream order */
            const int mode = (int)((modes >> (6 - 2 * kind)) & 3u);
            const int max_sym_k = kind == 0 ? 35 : (kind == 1 ? 31 : 52);
            const int max_log_k = kind == 1 ? 8 : 9;
            uint32_t *lt = kind == 0 ? L.ll : (kind == 1 ? L.of : L.ml);
            bmode[kind] = mode; bms[kind] = 0; blg[kind] = 0;
            if (mode == 0) { /* predefined: shared table built once per tile */
                ps.slot[kind] = kPredefSlot;
                ps.log[kind] = kind == 1 ? 5u : 6u;
            } else if (mode == 1) { /* RLE */
                if (left < 1u) return false;
                const uint32_t sy = uni(ip[0]);
                if ((int)sy > max_sym_k) return false;
                lt[0] = sy << 14;
                ip += 1; left -= 1u;
            } else if (mode == 2) {
                int ms = max_sym_k, lg = 0;
                const int used = read_ncount(L.norm3[kind], &ms, &lg, ip, left);
                if (used < 0 || lg > max_log_k) return false;
                bms[kind] = ms; blg[kind] = lg;
                ip += used; left -= (uint32_t)used;
            } else if (!ps.fse_valid) return false; /* repeat without a previous table */
        }
        __builtin_amdgcn_wave_barrier();
        PL_STAMP(18);
#pragma unroll
        for (int kind = 0; kind < 3; kind++) {
            if (bmode[kind] == 2) { /* the whole wave builds the table, straight into the workspace */
                const uint32_t goff = kind == 0 ? 0u : (kind == 1 ? 1024u : 512u);
                if (!fse_build_wave(L.cell3[kind], L.nxt3[kind], L.nxt, L.norm3[kind], bms[kind], blg[kind], gt + goff, kind, lane)) return false;
                ps.slot[kind] = k;
                ps.log[kind] = (uint32_t)blg[kind];
            } else if (bmode[kind] == 1) {
                const uint32_t *lt = kind == 0 ? L.ll : (kind == 1 ? L.of : L.ml);
                const uint32_t goff = kind == 0 ? 0u : (kind == 1 ? 1024u : 512u);
                copy_seq_table(gt + goff, lt, 1u, kind, lane);
                ps.slot[kind] = k;
                ps.log[kind] = 0u;
            }
        }
        PL_STAMP(19);
        __builtin_amdgcn_wave_barrier();
        PL_STAMP(63);
        ps.fse_valid = true;
        if ((uint32_t)(ip - L.sw) > 248u) return false; /* descriptions longer than the staged window: not a layout the libraries produce */
        d.sq_off = boff + used + (uint32_t)(ip - L.sw);
        d.sq_len = left;
        d.slots = ps.slot[0] | (ps.slot[1] << 8) | (ps.slot[2] << 16);
        d.logs = ps.log[0] | (ps.log[1] << 8) | (ps.log[2] << 16);
    }
    d.nseq = nseq;
    return true;
}

} // namespace

/* ------------------------------------------------------------------------------------------------ K1 */
__global__ void __launch_bounds__(64) k_zplan(ZPipe P)
{
    __shared__ __attribute__((aligned(16))) PlanLds L;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t f = blockIdx.x;

    if (f == 0u) { /* the predefined sequence tables, once per tile */
        for (int kind = 0; kind < 3; kind++) {
            int lg = 0;
            uint32_t *lt = kind == 0 ? L.ll : (kind == 1 ? L.of : L.ml);
            (void)read_seq_table(L, lt, &lg, kind, 0, nullptr, 0, false);
            __builtin_amdgcn_wave_barrier();
            copy_seq_table(P.predef + (kind == 0 ? 0u : (kind == 1 ? 1024u : 512u)), lt, 1u << lg, kind, lane);
            __builtin_amdgcn_wave_barrier();
        }
    }

    const uint64_t blk = P.first + f;
    const uint8_t *src = P.src_base + uni64(P.src_off[blk]);
    const uint32_t csize = uni(P.src_size[blk]);
    ZBlk *bd = P.blks + (uint64_t)f * P.nbmax;
    ZFrame fr = {};
    bool regular = false;
    uint32_t nblk = 0, total_seq = 0;
    do {
        if (csize < 5u) break;
        uint32_t magic;
        __builtin_memcpy(&magic, src, 4);
        if (uni(magic) != 0xFD2FB528u) break;
        stage256(L.hw, src, csize, lane);
        const uint8_t *fh = L.hw; /* the frame header is at most 18 bytes */
        const uint32_t fhd = uni(fh[4]);
        const uint32_t single = (fhd >> 5) & 1u, did = fhd & 3u, fcs_flag = fhd >> 6, has_ck = (fhd >> 2) & 1u;
        const uint32_t did_sz = did == 3u ? 4u : did;
        const uint32_t fcs_sz = fcs_flag == 0u ? single : (1u << fcs_flag);
        const uint32_t hsz = 5u + (single ? 0u : 1u) + did_sz + fcs_sz;
        if ((fhd & 0x08u) || csize < hsz) break;
        uint32_t p = 5u;
        if (!single) { if ((uni(fh[p]) >> 3) + 10u > 31u) break; p++; }
        if (did) {
            uint32_t id = 0;
            for (uint32_t k = 0; k < did_sz; k++) id |= uni(fh[p + k]) << (8u * k);
            if (id != 0u) break;
            p += did_sz;
        }
        uint64_t fcs = ~0ull;
        if (fcs_flag == 0u) { if (single) fcs = uni(fh[p]); }
        else {
            uint64_t v = 0;
            for (uint32_t k = 0; k < fcs_sz; k++) v |= (uint64_t)uni(fh[p + k]) << (8u * k);
            fcs = fcs_flag == 1u ? v + 256u : v;
        }
        uint32_t ip = hsz;
        PlanState ps = {};
        uint32_t lit_cursor = 0;
        bool okf = true;
        for (;;) {
            if (nblk >= P.nbmax || csize - ip < 3u) { okf = false; break; }
            stage256(L.hw, src + ip, csize - ip, lane); /* block header + the first 253 bytes of the block */
            const uint32_t bh = uni((uint32_t)L.hw[0] | ((uint32_t)L.hw[1] << 8) | ((uint32_t)L.hw[2] << 16));
            ip += 3u;
            const uint32_t last = bh & 1u, type = (bh >> 1) & 3u, bsize = bh >> 3;
            ZBlk d = {};
            d.type = type; d.src_off = ip; d.bsize = bsize;
            if (type == 3u) { okf = false; break; }
            if (type == 1u) {
                if (csize - ip < 1u) { okf = false; break; }
                ip += 1u;
            } else {
                if (bsize > csize - ip) { okf = false; break; }
                if (type == 2u) {
                    if (bsize >= kZBlockMax) { okf = false; break; }
                    if (!plan_block(L, P, ps, src + ip, L.hw + 3, bsize, ip, f, nblk, lit_cursor, d, lane)) { okf = false; break; }
                    total_seq += d.nseq;
                }
                ip += bsize;
            }
            if (lane == 0) bd[nblk] = d;
            nblk++;
            if (last) break;
        }
        if (!okf) break;
        if (has_ck) {
            if (csize - ip < 4u) break;
            fr.ck_off = ip;
            fr.flags |= F_CK;
            ip += 4u;
        }
        if (ip != csize) break; /* concatenated frames or trailing bytes: the fused decoder sorts it out */
        if (fcs != ~0ull) { fr.flags |= F_FCS; fr.fcs_lo = (uint32_t)fcs; fr.fcs_hi = (uint32_t)(fcs >> 32); }
        regular = true;
    } while (0);

    if (lane == 0) {
        if (regular && total_seq) {
            const uint32_t base = atomicAdd(&P.counters[0], total_seq);
            if ((uint64_t)base + total_seq > P.seqcap) regular = false;
            else {
                uint32_t b = base;
                for (uint32_t k = 0; k < nblk; k++) { bd[k].seq_base = b; b += bd[k].nseq; }
            }
        }
        if (regular) {
            for (uint32_t k = 0; k < nblk; k++)
                if (bd[k].type == 2u && bd[k].lit_mode == 2u) {
                    P.hitems[atomicAdd(&P.counters[1], 1u)] = f * P.nbmax + k;
                    atomicAdd(&P.counters[4 + (bd[k].huf_log & 15u)], 1u); /* histogram of table logs (diagnostics) */
                }
            for (uint32_t k = 0; k < nblk; k++)
                if (bd[k].type == 2u && bd[k].nseq) P.sitems[atomicAdd(&P.counters[3], 1u)] = f * P.nbmax + k;
            for (uint32_t k = 0; k < nblk; k++)
                if (bd[k].type == 2u && bd[k].nseq) {
                    atomicAdd(&P.counters[20 + (bd[k].logs & 15u)], 1u);
                    atomicAdd(&P.counters[32 + ((bd[k].logs >> 8) & 15u)], 1u);
                    atomicAdd(&P.counters[44 + ((bd[k].logs >> 16) & 15u)], 1u);
                }
        } else {
            fr.flags = F_IRREG;
            P.irregular[atomicAdd(&P.counters[2], 1u)] = f;
        }
        fr.nblk = nblk;
        P.frames[f] = fr;
    }
}

/* ------------------------------------------------------------------------------------------------ K2 */
#ifndef CRYO_ZHUF_PER_WAVE
#define CRYO_ZHUF_PER_WAVE 16
#endif
constexpr uint32_t kHufPerWave = CRYO_ZHUF_PER_WAVE; /* blocks per wave: 16 x 4 streams = 64 lanes */
constexpr uint32_t kHufL1 = 11;      /* table in LDS: 2^11 entries (4 KiB) per block -- every table libzstd's encoder emits */

constexpr uint32_t kHufL2 = 128;     /* second-level entries per block */

/* LDS capacity bounds this kernel (streams in flight per CU = LDS / table bytes per stream; the lookup
 * latency per symbol is fixed).  A table of 2^12 entries (legal, never produced by libzstd) is kept as two
 * levels: the first indexed by the next 11 bits; prefixes under which 12-bit codes live carry nbits = 0 and
 * the number of a 2-entry sub-table (255: no room, resolved in the full table in global memory).
 * A two-level table with a 9-bit first level was measured: 2.5x the waves per CU but twice the time per
 * symbol (the escape test serialises each lookup), a net loss at 11 bits. */
struct HufLds {
    uint16_t tbl[kHufPerWave << kHufL1];
    uint16_t sub[kHufPerWave * kHufL2];
    uint32_t ring[36][64];
};

struct HufTab {
    const uint16_t *t, *t2, *gt;
    uint32_t l1, hlog, sh;
};

template <bool TWO>
__device__ inline uint32_t huf_symbol(LaneBits<64> &lb, const HufTab &h)
{
    uint32_t e = h.t[lb.peek_nz(h.l1)];
    if (TWO && (e >> 8) == 0u) { /* long code */
        const uint32_t full = lb.peek_nz(h.hlog), s = e & 255u;
        if (s != 255u) e = h.t2[(s << h.sh) | (full & ((1u << h.sh) - 1u))];
        else {
            e = h.gt[full];
            asm volatile("" : "+v"(e)); /* take the load's wait here, not at the join every lane passes */
        }
    }
    lb.skip(e >> 8);
    return e & 255u;
}

/* eight symbols -> one 8-byte store; two container fills (4 x 12 bits <= 57) */
template <int J, bool TWO>
__device__ inline void huf_octet(LaneBits<64> &lb, const HufTab &h, uint8_t *o, uint32_t r, uint32_t n8)
{
    lb.tick<J>();
    if (r + J < n8) {
        uint64_t acc = 0;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            lb.fill();
#pragma unroll
            for (int q = 0; q < 4; q++) acc |= (uint64_t)huf_symbol<TWO>(lb, h) << (8 * (4 * half + q));
        }
        __builtin_memcpy(o + (size_t)(r + J) * 8u, &acc, 8);
    }
}

__global__ void __launch_bounds__(64) k_zhuf(ZPipe P, const uint32_t *__restrict__ items, const uint32_t counter)
{
    __shared__ __attribute__((aligned(16))) HufLds L;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nitems = uni(P.counters[counter]);
    /* grid-stride (round 4): as the walkers' fallback this kernel usually has nothing to do, and 2 304 workgroups that each
     * need 70 KB of LDS before they can find that out took 0.7 ms per tile to come and go, in the tile's stream, in front of
     * k_zchain; the launcher sizes the grid for the chip (two of these fit a CU), not for the worst case */
    for (uint32_t base = blockIdx.x * kHufPerWave; base < nitems; base += gridDim.x * kHufPerWave) {
    if (base != blockIdx.x * kHufPerWave) { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }

    /* stage the 16 first-level tables */
    for (uint32_t j = 0; j < kHufPerWave && base + j < nitems; j++) {
        const uint32_t it = uni(items[base + j]);
        const ZBlk *d = P.blks + it;
        const uint32_t f = it / P.nbmax;
        const uint32_t hlog = uni(d->huf_log);
        const uint16_t *g = P.huf + ((uint64_t)f * P.nbmax + uni(d->huf_slot)) * kHufTblWords;
        uint16_t *l = L.tbl + (j << kHufL1);
        if (hlog <= kHufL1) {
            for (uint32_t i = lane; i < (1u << hlog); i += 64u) l[i] = g[i];
        } else {
            const uint32_t sh = hlog - kHufL1, cap = kHufL2 >> sh;
            uint16_t *l2 = L.sub + j * kHufL2;
            uint32_t running = 0;
            for (uint32_t i = lane; i < (1u << kHufL1); i += 64u) {
                uint32_t e = g[i << sh];
                const bool lng = (e >> 8) > kHufL1;
                const unsigned long long m = wave_ballot(lng);
                if (lng) {
                    const uint32_t rank = running + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (rank < cap) {
                        for (uint32_t k = 0; k < (1u << sh); k++) l2[(rank << sh) + k] = g[(i << sh) + k];
                        e = rank;
                    } else e = 255u;
                }
                l[i] = (uint16_t)e;
                running += (uint32_t)__builtin_popcountll(m);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    const uint32_t j = lane >> 2, sid = lane & 3u;
    bool have = j < kHufPerWave && base + j < nitems;
    const uint32_t it = have ? items[base + j] : items[base];
    const ZBlk *d = P.blks + it;
    const uint32_t f = it / P.nbmax;
    const uint32_t nstreams = d->nstreams;
    const uint32_t jt = have ? j : 0u; /* lanes without a block of their own look at the wave's first table */
    have = have && sid < nstreams;
    const uint32_t regen = d->regen, hlog = d->huf_log;
    const uint32_t l1 = hlog < kHufL1 ? hlog : kHufL1;
    const uint32_t seg = (regen + 3u) / 4u;
    uint32_t cnt, oofs;
    if (nstreams == 1u) { cnt = regen; oofs = 0; }
    else { cnt = sid < 3u ? seg : regen - 3u * seg; oofs = sid * seg; }
    const uint64_t blk = P.first + f;
    const uint8_t *src = P.src_base + P.src_off[blk];
    uint8_t *o = P.lits + (uint64_t)f * P.litcap + d->lit_src + oofs;
    const uint32_t soff = d->hs_off[sid], slen = d->hs_len[sid];
    const uint16_t *gt = P.huf + ((uint64_t)f * P.nbmax + d->huf_slot) * kHufTblWords;

    bool ok = true;
    LaneBits<64> lb;
    const bool opened = lb.init(&L.ring[0][lane], src, soff, slen, have);
    if (have && !opened) ok = false;
    HufTab h;
    h.t = L.tbl + (jt << kHufL1);
    h.t2 = L.sub + jt * kHufL2;
    h.gt = gt;
    h.l1 = l1; h.hlog = hlog; h.sh = hlog > kHufL1 ? hlog - kHufL1 : 0u;
    const uint32_t n8 = opened ? cnt >> 3 : 0u;
    const uint32_t maxn8 = wave_max(n8);
    if (wave_max(hlog) <= kHufL1) {
        for (uint32_t r = 0; r < maxn8; r += 4u) {
            huf_octet<0, false>(lb, h, o, r, n8);
            huf_octet<1, false>(lb, h, o, r, n8);
            huf_octet<2, false>(lb, h, o, r, n8);
            huf_octet<3, false>(lb, h, o, r, n8);
        }
    } else {
        for (uint32_t r = 0; r < maxn8; r += 4u) {
            huf_octet<0, true>(lb, h, o, r, n8);
            huf_octet<1, true>(lb, h, o, r, n8);
            huf_octet<2, true>(lb, h, o, r, n8);
            huf_octet<3, true>(lb, h, o, r, n8);
        }
    }
    if (opened) { /* the last cnt % 8 symbols */
        uint32_t i = n8 * 8u;
        lb.fill();
        for (uint32_t q = 0; q < 4u && i < cnt; q++, i++) o[i] = (uint8_t)huf_symbol<true>(lb, h);
        lb.fill();
        for (; i < cnt; i++) o[i] = (uint8_t)huf_symbol<true>(lb, h);
        ok = lb.pos == 0; /* must end exactly */
    }
    if (have && !ok) atomicOr(&P.frames[f].flags, F_BAD);
    }
}

/* per-lane input rings fed cooperatively by the wave (k_zhufw, k_zchain) */
/* waves per SIMD the register allocator aims at for k_zexec: 5 (96 registers).  Round 4 ran it at 6 (80 registers, 32 bytes of
 * scratch: 1 % on the call); with the segmented literal stream of round 6 (Wave::fetch_seg keeps a piece per lane) 6 waves mean
 * 108 bytes of scratch and 5 are faster on every shape (profiles/r06_zstd_decode.txt) */
constexpr int kZexecOcc = 5;
constexpr uint32_t kChRing = 128, kChStride = 144; /* ring + 8-byte mirror + pad */
__device__ inline uint32_t bperm32(uint32_t v, uint32_t src_lane)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}

/* ------------------------------------------------------------------------------------------------ K2' */
/* k_zhufw: a WAVE per block, 16 walkers per Huffman stream.
 *
 * k_zhuf keeps 16 blocks' tables in LDS for 64 lanes: 4 KiB of table per 4 lanes, two waves per CU, every symbol a
 * dependent LDS lookup on a lone wave.  A prefix code resynchronises: a decoder started at a wrong bit position
 * lands on a true symbol boundary after a few symbols and is identical to the true decoder from there on.  So here a
 * stream's bits are cut into up to 16 segments of equal length; walker w starts at the GUESSED position (the first bit
 * of segment w), decodes its segment and keeps going into segment w+1 until it stands on a position walker w+1 stood on
 * too -- every walker notes where it stands (and how many symbols it has) after every fourth of its first 60 turns, and
 * its left neighbour, once its chain has merged, passes through all of these.  By induction from walker 0, whose start
 * is true, walker w+1's symbols are true from that mark on.  Symbols go to a scratch region per walker (their final
 * position is not known before the walkers to the left are counted) and are moved to the literal pool by the wave at
 * the end.  One table serves 64 lanes, seven waves fit a CU, and the kernel is bound by instruction issue and bandwidth
 * instead of LDS capacity.
 *
 * How fast chains merge depends on the code: with lengths spread over 3 .. 9 bits (text) a chain merges in ~5 symbols;
 * where nearly all symbols have ONE length (literals that are hex digits: 14 codes of 4 bits) two chains at different
 * offsets modulo that length stay apart until a rare longer code comes by, ~80 symbols on average, so the walk into the
 * next segment takes whole turns like the body (ext_octet) and the marks cover ~4 000 bits.  (First version: a bitmap of
 * the symbol starts in the segment's first 512 bits and a symbol-by-symbol walk: 61 % of the blocks of 1 MiB frames of
 * hex-heavy rows had a boundary that did not merge inside the window and were handed back.)
 *
 * A walker that runs out of its neighbour's marks knows where the neighbour's segment truly starts: the neighbour
 * decodes its segment once more from there (the repair pass).  Anything else unusual -- a 12-bit table, two such misses
 * in a row, a scratch region that
 * overflows (symbol density more than twice the stream's average), counts that do not add up, a stream that does not
 * end on its first bit -- puts the block on a second list that k_zhuf decodes afterwards, lane per stream as before:
 * verdicts are k_zhuf's. */
constexpr uint32_t kHwMarks = 16;  /* positions a walker notes for its left neighbour: its start and where it stands after every fourth of its first 60 turns */
constexpr uint32_t kHwSlack = 400; /* scratch bytes per walker beyond twice the average */
constexpr uint32_t kHwBlockSlack = 64u * kHwSlack + 128u; /* + alignment: no two blocks' scratch in one 128-byte line */

constexpr uint32_t kHwRing = 128, kHwStride = 144; /* four 32-byte units per walker; 16 bytes per turn arrive, eight lookups eat at most 11 */

/* decoding table, two symbols per lookup where the second one's code fits behind the first in the 11 bits looked at
 * (libzstd's "X2" idea): symbol 1 | symbol 2 << 8 (0 if none) | length 1 << 16 | bits of both << 20 | symbols - 1 << 24 */
struct HufwLds {
    uint32_t tbl[1u << kHufL1];
    uint8_t ring[64 * kHwStride + 64 * 16];
    uint32_t hp[65][kHwMarks]; /* bits below the walker's start (0xFFFF: no mark) | symbols the walker had there << 16 */
    uint32_t skip[65];
};

struct HwLane {
    int32_t s0, pos, cb, lowh, fillh;
    uint32_t pend;
};

/* the ring feed of k_zchain for 64 walkers of ONE frame, in 32-byte units: turn J serves walkers 32 (J & 1) .. + 31, two
 * lanes x 16 bytes each; lowh / fillh count units here */
template <int J>
__device__ inline void hw_feed(uint8_t *ring, HwLane &z, const uint32_t lane, const bool alive, uint4 &fd, uint32_t &fa, uint32_t &fm,
                               const uint8_t *gsrc, const uint32_t vend)
{
    *reinterpret_cast<uint4 *>(ring + fa) = fd;
    *reinterpret_cast<uint2 *>(ring + fm) = make_uint2(fd.x, fd.y);
    const uint32_t got = (z.pend >> J) & 1u;
    z.fillh -= (int32_t)got;
    z.pend &= ~(1u << J);
    /* room: the unit this one replaces in the ring lies wholly above the reader */
    const bool want = alive & ((lane >> 5) == (uint32_t)(J & 1)) & (z.cb < 32 * (z.lowh + (int32_t)(kHwRing / 32u) - 1)) & (32 * z.lowh > z.s0);
    const uint32_t wi = want ? 1u : 0u;
    z.lowh -= (int32_t)wi;
    z.pend |= wi << J;
    const uint32_t srv = 32u * (J & 1) + (lane >> 1);
    const uint32_t m = bperm32((uint32_t)z.lowh | (wi << 31), srv);
    const bool p = (m >> 31) != 0u;
    const uint32_t o = ((m & 0x7FFFFFFFu) << 5) + (lane & 1u) * 16u;
    const uint32_t tr = 64u * kHwStride + lane * 16u;
    fa = p ? srv * kHwStride + (o & (kHwRing - 1u)) : tr;
    fm = (p & ((o & (kHwRing - 1u)) == 0u)) ? srv * kHwStride + kHwRing : tr;
    fd = *reinterpret_cast<const uint4 *>(gsrc + ((p & (o < vend)) ? o : 0u));
}

__device__ inline bool hw_ready(const HwLane &z) { return (z.cb - 16 >= 32 * z.fillh) | (32 * z.fillh <= z.s0); }

__device__ inline uint64_t hw_window(const uint8_t *ring, const HwLane &z, const uint32_t myring)
{
    uint64_t r;
    __builtin_memcpy(&r, ring + myring + ((uint32_t)(z.cb - 7) & (kHwRing - 1u)), 8);
    return r << (7u - ((uint32_t)(z.pos - 1) & 7u));
}

/* the first n (< 16) bytes of v */
__device__ inline void store_head(uint8_t *p, uint4 v, uint32_t n)
{
    if (n & 8u) { __builtin_memcpy(p, &v.x, 8); p += 8; v.x = v.z; v.y = v.w; }
    if (n & 4u) { __builtin_memcpy(p, &v.x, 4); p += 4; v.x = v.y; }
    if (n & 2u) { const uint16_t h = (uint16_t)v.x; __builtin_memcpy(p, &h, 2); p += 2; v.x >>= 16; }
    if (n & 1u) *p = (uint8_t)v.x;
}

__device__ __attribute__((always_inline)) inline void zhufw_item(const ZPipe &P, HufwLds &L, const uint32_t item)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t it = uni(P.hitems[item]);
    const ZBlk *d = P.blks + it;
    const uint32_t f = it / P.nbmax, kblk = it - f * P.nbmax;
    const uint32_t hlog = uni(d->huf_log), nstreams = uni(d->nstreams), regen = uni(d->regen);
#ifdef CRYO_HW_PROF
    auto fallback = [&](int why = 0) { if (lane == 0u) { P.hitems2[atomicAdd(&P.counters[61], 1u)] = it; atomicAdd(&P.counters[why == 0 ? 30 : (why == 1 ? 31 : (why == 2 ? 42 : (why == 3 ? 43 : (why == 4 ? 54 : 55))))], 1u); } };
#else
    auto fallback = [&](int = 0) { if (lane == 0u) P.hitems2[atomicAdd(&P.counters[61], 1u)] = it; };
#endif
    if (hlog > kHufL1 || hlog == 0u || regen < P.hufw_min) { fallback(0); return; }
    {
        /* the one-symbol table goes through the (still unused) rings */
        const uint16_t *g = P.huf + ((uint64_t)f * P.nbmax + uni(d->huf_slot)) * kHufTblWords;
        uint16_t *x1 = reinterpret_cast<uint16_t *>(L.ring);
        for (uint32_t i = lane; i < (1u << hlog); i += 64u) x1[i] = g[i];
        for (uint32_t i = lane; i < 65u * kHwMarks; i += 64u) (&L.hp[0][0])[i] = (i % kHwMarks) == 0u ? 0u : 0xFFFFu;
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        const uint32_t dsh = kHufL1 - hlog;
        for (uint32_t i = lane; i < (1u << kHufL1); i += 64u) {
            const uint32_t e1 = x1[i >> dsh], l1 = e1 >> 8;
            const uint32_t e2 = x1[((i << l1) & ((1u << kHufL1) - 1u)) >> dsh], l2 = e2 >> 8;
            const bool two = l1 + l2 <= kHufL1;
            L.tbl[i] = (e1 & 255u) | (two ? (e2 & 255u) << 8 : 0u) | (l1 << 16) | ((two ? l1 + l2 : l1) << 20) | ((two ? 1u : 0u) << 24);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    }
#ifdef CRYO_HW_PROF
    uint64_t tprev = __builtin_amdgcn_s_memtime();
#define HW_STAMP(k) do { const uint64_t tn = __builtin_amdgcn_s_memtime(); if (lane == 0u) atomicAdd(&P.counters[k], (uint32_t)((tn - tprev) >> 6)); tprev = tn; } while (0)
#else
#define HW_STAMP(k) do { } while (0)
#endif
    const uint32_t sid = lane >> 4, w = lane & 15u;
    const uint32_t seg = (regen + 3u) / 4u;
    uint32_t cnt, oofs;
    if (nstreams == 1u) { cnt = regen; oofs = 0; }
    else { cnt = sid < 3u ? seg : regen - 3u * seg; oofs = sid * seg; }
    const uint64_t fo = uni64(P.src_off[P.first + f]);
    const uint64_t aoff = fo & ~(uint64_t)63;
    const uint32_t delta = (uint32_t)(fo & 63u);
    const uint32_t vend = delta + uni(P.src_size[P.first + f]);
    const uint8_t *gsrc = P.src_base + aoff;
    const uint32_t soff = d->hs_off[sid < nstreams ? sid : 0u], slen = d->hs_len[sid < nstreams ? sid : 0u];
    bool sok = sid < nstreams && slen >= 1u;
    uint32_t last = 0;
    if (sok) last = gsrc[delta + soff + slen - 1u];
    if (last == 0u) sok = false;
    if (wave_any(sid < nstreams && !sok)) { fallback(1); return; } /* a stream without an end mark: k_zhuf says what it is */
    /* segments */
    const uint32_t T = sok ? (slen - 1u) * 8u + (31u - (uint32_t)__builtin_clz(last)) : 0u;
    uint32_t se = T >> P.hufw_seglog;
    se = se < 1u ? 1u : (se > 16u ? 16u : se);
    const uint32_t Lb = (T + se - 1u) / se;
    const bool walker = sok && w < se;
    const bool lastw = w + 1u == se;
    const int32_t Pw = (int32_t)T - (int32_t)(w * Lb);
    const int32_t bound = lastw ? 0 : Pw - (int32_t)Lb; /* first bit position of the next segment */
    /* scratch: twice the average count + slack per walker */
    const uint32_t avg = (cnt + se - 1u) / se;
    const uint32_t cap = ((2u * avg + 15u) & ~15u) + kHwSlack - 48u;
    uint8_t *const tbase = P.htmp + (uint64_t)f * P.htmp_stride + ((2u * (uint64_t)uni(d->lit_src) + 127u) & ~(uint64_t)127) + (uint64_t)kblk * kHwBlockSlack;
    uint8_t *const tmp = tbase + 2u * oofs + sid * 16u * kHwSlack + w * (((2u * avg + 15u) & ~15u) + kHwSlack - 32u);
    const uint32_t myring = lane * kHwStride;
    HwLane z;
    z.s0 = (int32_t)(delta + soff);
    /* the ring's four units at and below the walker's first byte, loaded by the lane itself */
    auto prime = [&](const bool who, const int32_t from) {
        if (!who) return;
        z.pos = from;
        z.cb = z.s0 + ((z.pos - 1) >> 3);
        z.pend = 0;
        const int32_t ht = z.cb >> 5;
        const int32_t hl = ht >= 3 ? ht - 3 : 0;
        for (int32_t h = hl; h <= ht; h++)
            for (uint32_t q = 0; q < 2u; q++) {
                const uint32_t o = (uint32_t)h * 32u + q * 16u;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (o < vend) v = *reinterpret_cast<const uint4 *>(gsrc + o);
                *reinterpret_cast<uint4 *>(L.ring + myring + (o & (kHwRing - 1u))) = v;
                if ((o & (kHwRing - 1u)) == 0u) *reinterpret_cast<uint2 *>(L.ring + myring + kHwRing) = make_uint2(v.x, v.y);
            }
        z.lowh = z.fillh = hl;
    };
    z.pos = 0; z.cb = z.s0 - 1; z.pend = 0; z.lowh = z.fillh = 0;
    prime(walker, Pw);
    __builtin_amdgcn_wave_barrier();

    enum { MAIN = 0, EXT = 1, DONE = 2 };
    uint32_t phase = walker ? MAIN : DONE;
    bool okw = true;
    uint32_t n = 0, sync_j = 0, hc = 1, ej = 0, tc = 0, main_n = 0;
    int32_t e_pos = 0;   /* where the walker left its segment */
    bool nomeet = false; /* ... and did not meet its right neighbour before that one's marks ended */
    const uint32_t tr = 64u * kHwStride + lane * 16u;
    uint4 fd0 = make_uint4(0, 0, 0, 0), fd1 = fd0, fd2 = fd0, fd3 = fd0;
    uint32_t fa0 = tr, fa1 = tr, fa2 = tr, fa3 = tr, fm0 = tr, fm1 = tr, fm2 = tr, fm3 = tr;

    /* one symbol, every check (the tail of a segment and the walk into the next one) */
    auto slow = [&]() __attribute__((always_inline)) {
        if (phase != DONE && hw_ready(z)) {
            if (phase == MAIN && z.pos <= bound) { /* the symbol that starts here belongs to the next segment */
                if (lastw) { phase = DONE; okw = z.pos == 0; }
                else { phase = EXT; main_n = n; e_pos = z.pos; }
            }
            if (phase == EXT) { /* does the right neighbour stand here after one of its first turns? */
                const uint32_t dd = (uint32_t)(bound - z.pos);
                uint32_t lim = L.hp[lane + 1u][ej] & 0xFFFFu;
                if (lim < dd && ej + 1u < kHwMarks) lim = L.hp[lane + 1u][++ej] & 0xFFFFu;
                if (lim == dd) { phase = DONE; sync_j = ej; }
                else if (lim < dd || lim == 0xFFFFu) { phase = DONE; okw = false; nomeet = true; } /* behind its last mark */
            }
            if (phase != DONE) {
                const uint64_t c = hw_window(L.ring, z, myring);
                const uint32_t e = L.tbl[(uint32_t)(c >> 32) >> (32u - kHufL1)];
                z.pos -= (int32_t)((e >> 16) & 15u);
                z.cb = z.s0 + ((z.pos - 1) >> 3);
                if (n < cap && z.pos >= 0 && ((e >> 16) & 15u) != 0u) { tmp[n] = (uint8_t)e; n++; }
                else { phase = DONE; okw = false; }
            }
        }
    };
    /* eight lookups = 8 .. 16 symbols, while more than 88 bits of the segment are left.  Each half stores 8 bytes of which
     * 4 .. 8 are symbols: the next store overwrites the rest.  After each of the first turns the walker notes where it
     * stands and how many symbols it has: its left neighbour, walking symbol by symbol, will stand on one of these. */
    auto octet = [&](const bool mark) __attribute__((always_inline)) {
        if (phase == MAIN && z.pos - bound > 88 && hw_ready(z)) {
            if (n + 24u > cap) { phase = DONE; okw = false; }
            else {
                /* ONE 16-byte store per turn (two halves of 4 .. 8 symbols each).  The walkers' stores are scattered -- 64 lines
                 * per instruction -- and it is the number of such requests, not their bytes, that bounds this kernel: with an
                 * 8-byte store per half a tile took 7.0 ms, with this 5.9 (profiles/r06_zstd_decode.txt; the input ring
                 * replaced by a window in registers, twelve waves per CU instead of seven, but a load per lane and turn:
                 * 10.4 ms, profiles/scripts/r06_zhufw_regwin.patch) */
                uint64_t acc2[2];
                uint32_t k2[2];
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    uint64_t c = hw_window(L.ring, z, myring);
                    uint64_t acc = 0;
                    uint32_t k = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t e = L.tbl[(uint32_t)(c >> 32) >> (32u - kHufL1)];
                        const uint32_t lt = (e >> 20) & 15u;
                        c <<= lt;
                        z.pos -= (int32_t)lt;
                        acc |= (uint64_t)(e & 0xFFFFu) << (8u * k);
                        k += 1u + (e >> 24);
                    }
                    z.cb = z.s0 + ((z.pos - 1) >> 3);
                    acc2[half] = acc;
                    k2[half] = k;
                }
                {
                    const uint32_t sh = 8u * k2[0]; /* 32 .. 64; bytes k .. 7 of a half's word are zero */
                    const uint64_t lo = acc2[0] | (sh < 64u ? acc2[1] << sh : 0ull);
                    const uint64_t hi = sh < 64u ? acc2[1] >> (64u - sh) : acc2[1];
                    const uint4 v = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
                    __builtin_memcpy(tmp + n, &v, 16);
                    n += k2[0] + k2[1];
                }
                if (mark) { /* every fourth turn: 15 marks over ~4 000 bits */
                    tc++;
                    if ((tc & 3u) == 0u && hc < kHwMarks) {
                        L.hp[lane][hc] = (uint32_t)(Pw - z.pos) | (n << 16);
                        hc++;
                    }
                }
            }
        }
    };
    /* the same on the walk into the next segment: before every symbol, is this where the right neighbour stood after one of
     * its turns?  (Symbol by symbol -- slow() -- this walk was longer than the segment's body on data whose codes are
     * nearly all of one length, hex digits: chains at different offsets modulo that length merge once in ~80 symbols.) */
    auto ext_octet = [&]() __attribute__((always_inline)) {
        if (n + 24u > cap) { phase = DONE; okw = false; return; }
        bool live = true;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint32_t lim = L.hp[lane + 1u][ej] & 0xFFFFu;
            if (live && lim < (uint32_t)(bound - z.pos) && ej + 1u < kHwMarks) lim = L.hp[lane + 1u][++ej] & 0xFFFFu; /* marks are 32 lookups apart: one per half at most */
            uint64_t c = hw_window(L.ring, z, myring);
            uint64_t acc = 0;
            uint32_t k = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t e = L.tbl[(uint32_t)(c >> 32) >> (32u - kHufL1)];
                const uint32_t l1 = (e >> 16) & 15u, lt = (e >> 20) & 15u, two = e >> 24;
                const uint32_t dd = (uint32_t)(bound - z.pos);
                const bool hit1 = live & (lim == dd);
                const bool hit2 = live & !hit1 & (two != 0u) & (lim == dd + l1);
                acc |= (uint64_t)(e & 0xFFFFu) << (8u * k);
                k += (live & !hit1) ? (hit2 ? 1u : 1u + two) : 0u;
                if (hit1 | hit2) { live = false; sync_j = ej; phase = DONE; }
                c <<= lt;
                z.pos -= live ? (int32_t)lt : 0;
            }
            z.cb = z.s0 + ((z.pos - 1) >> 3);
            __builtin_memcpy(tmp + n, &acc, 8);
            n += k;
            if (live && (lim == 0xFFFFu || (lim < (uint32_t)(bound - z.pos) && ej + 1u >= kHwMarks))) { live = false; phase = DONE; okw = false; nomeet = true; } /* behind its last mark */
        }
    };
    auto tail_step = [&]() __attribute__((always_inline)) {
        if (phase == EXT && z.pos > 88 && hw_ready(z)) ext_octet();
        else slow();
    };
#define HW_TURNS(BODY) \
        hw_feed<0>(L.ring, z, lane, phase != DONE, fd0, fa0, fm0, gsrc, vend); BODY; \
        hw_feed<1>(L.ring, z, lane, phase != DONE, fd1, fa1, fm1, gsrc, vend); BODY; \
        hw_feed<2>(L.ring, z, lane, phase != DONE, fd2, fa2, fm2, gsrc, vend); BODY; \
        hw_feed<3>(L.ring, z, lane, phase != DONE, fd3, fa3, fm3, gsrc, vend); BODY;
    HW_STAMP(56);
#define HW_RUN() \
    /* head: the turns that leave marks; body: the rest */ \
    for (uint32_t t = 0; t < 4u * kHwMarks && wave_any(phase == MAIN && z.pos - bound > 88); t += 4u) { HW_TURNS(octet(true)) } \
    while (wave_any(phase == MAIN && z.pos - bound > 88)) { HW_TURNS(octet(false)) } \
    /* tail: the rest of the segment (at most 88 bits each: symbol by symbol), then on into the next one until the chains meet */ \
    /* (round 4 tried these four loops in their rotated form, test at the bottom, which rid lane_runs of its register copies: \
     * here the body shrank by 8 % and the kernel took 9 % longer, 3.17 -> 3.45 ms per tile; profiles/r04_valu.txt) */ \
    while (wave_any(phase == MAIN)) { HW_TURNS(if (phase == MAIN) slow()) } \
    while (wave_any(phase != DONE)) { HW_TURNS(tail_step()) }
    HW_RUN()
    HW_STAMP(58);
    /* ---- repair: a walker that did not meet its right neighbour in time knows where that one's segment truly starts
     * (where it left its own): the neighbour decodes its segment once more from there, alone; the walker keeps the
     * symbols of its own segment only.  (1 boundary in 7 000 on literals of one code length; a second failure in a row
     * hands the block back.) ---- */
    bool redone = false;
    uint32_t ep_used = 0;
    {
        const bool give = walker && !lastw && !okw && nomeet;
        if (give) { n = main_n; okw = true; }
        L.skip[lane + 1u] = give ? (uint32_t)e_pos : 0xFFFFFFFFu;
        if (w == 0u) L.skip[lane] = 0xFFFFFFFFu;
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        const uint32_t ep = (walker && w != 0u) ? L.skip[lane] : 0xFFFFFFFFu;
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        redone = ep != 0xFFFFFFFFu;
        ep_used = ep;
        if (wave_any(redone)) {
            fa0 = fa1 = fa2 = fa3 = fm0 = fm1 = fm2 = fm3 = tr; /* pieces on their way belong to the old positions */
            prime(redone, (int32_t)ep);
            if (redone) { phase = MAIN; okw = true; nomeet = false; n = 0; sync_j = 0; ej = 0; hc = kHwMarks; main_n = 0; }
            __builtin_amdgcn_wave_barrier();
            HW_RUN()
        }
    }
    HW_STAMP(59);
#undef HW_RUN
#undef HW_TURNS
    /* ---- who is true from where: walker w tells walker w+1 how many of its symbols lie before the meeting point ---- */
    const bool right_redone = __shfl((int)redone, (int)((lane + 1u) & 63u), 64) != 0 && !lastw;
    if (walker && right_redone) n = main_n; /* the right neighbour decoded its segment again from where this one left its own */
    /* ... which must still be that place (it is not, if this walker was decoded again itself and came out elsewhere) */
    const uint32_t left_e = (uint32_t)__shfl((int)e_pos, (int)((lane - 1u) & 63u), 64); /* by every lane: a shuffle inside `redone && ...` reads lanes that are switched off */
    const bool stale = redone && left_e != ep_used;
    {
        uint32_t sk = 0;
        if (walker && !lastw && okw && !right_redone) sk = L.hp[lane + 1u][sync_j] >> 16;
        L.skip[lane + 1u] = sk;
        if (w == 0u) L.skip[lane] = 0u; /* lane 16 k is also written by lane 16 k - 1 (a last walker: 0) */
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const uint32_t myskip = (walker && w != 0u && !redone) ? L.skip[lane] : 0u;
    const uint32_t ntrue = walker ? n - myskip : 0u;
    const uint32_t incl = scan16_incl(ntrue);
    const uint32_t total = (uint32_t)__shfl((int)incl, (int)(lane | 15u), 64);
    const bool fine = (!walker || (okw && n >= myskip && !stale)) && (sid >= nstreams || total == cnt);
    if (!wave_all(fine)) { fallback(wave_any(walker && !okw && phase == DONE && n + 24u > cap) ? 4 : (wave_any(walker && !okw) ? 2 : (wave_any(walker && n < myskip) ? 5 : 3))); return; }
    /* ---- the move of the walkers' symbols to the literal pool is k_zmove's (round 4): here it was 35 % of this
     * kernel's wave time at seven waves per CU (22 KB of LDS per wave); a kernel that does nothing else runs it with the
     * chip full of waves ---- */
    {
        const uint32_t srcp = (uint32_t)(tmp - tbase) + myskip;
        const uint32_t dstp = oofs + (incl - ntrue);
        P.hsegs[(uint64_t)it * 64u + lane] = make_uint4(srcp, ntrue, dstp, 0u);
        /* read in place by k_zexec (Wave::fetch_seg) when every piece that is not empty has 8 bytes; moved to the pool otherwise */
        const bool in_place = P.in_place != 0u && wave_all(ntrue == 0u || ntrue >= 8u);
        if (lane == 0u) {
            if (in_place) P.blks[it].segd = 1u;
            else P.mitems[atomicAdd(&P.counters[60], 1u)] = it;
        }
    }
    HW_STAMP(62);
}

/* grid-stride over the tile's Huffman blocks (round 4): the grid used to be one workgroup per descriptor SLOT (frames x
 * blocks a frame may have: three times the blocks there are at 128 KiB), and a workgroup needs its 22 KB of LDS before it
 * can find out that it has no block */
__global__ void __launch_bounds__(64) k_zhufw(ZPipe P)
{
    __shared__ __attribute__((aligned(16))) HufwLds L;
    const uint32_t n = uni(P.counters[1]);
    for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
        if (item != blockIdx.x) { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }
        zhufw_item(P, L, item);
    }
}

/* k_zmove: a wave per block k_zhufw decoded: its walkers' true symbols, scratch -> literal pool.  Whole 16-byte stores,
 * segment after segment in the order of their destinations: what a segment's last store writes beyond its symbols is
 * overwritten by the next segment's first store (same wave: in order), and behind a block's literals the pool has 16
 * spare bytes.  Four segments' loads are issued before the previous four are stored. */
__global__ void __launch_bounds__(64) k_zmove(ZPipe P)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (blockIdx.x >= uni(P.counters[60])) return;
    const uint32_t it = uni(P.mitems[blockIdx.x]);
    const ZBlk *d = P.blks + it;
    const uint32_t f = it / P.nbmax, kblk = it - f * P.nbmax;
    const uint8_t *const tbase = P.htmp + (uint64_t)f * P.htmp_stride + ((2u * (uint64_t)uni(d->lit_src) + 127u) & ~(uint64_t)127) + (uint64_t)kblk * kHwBlockSlack;
    uint8_t *const lit = P.lits + (uint64_t)f * P.litcap + uni(d->lit_src);
    const uint4 sg4 = P.hsegs[(uint64_t)it * 64u + lane];
    const uint32_t srcp = sg4.x, ntrue = sg4.y, dstp = sg4.z;
    const uint32_t o0 = lane * 16u;
    uint4 va[4], vb[4], wa[4], wb[4];
    auto loads = [&](uint4 (&xa)[4], uint4 (&xb)[4], const uint32_t sg) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t nn = lane_get(ntrue, sg + u);
            const uint8_t *sp = tbase + lane_get(srcp, sg + u);
            xa[u] = xb[u] = make_uint4(0, 0, 0, 0);
            if (o0 < nn) __builtin_memcpy(&xa[u], sp + o0, 16); /* up to 15 bytes beyond the symbols: inside the walker's region */
            if (o0 + 1024u < nn) __builtin_memcpy(&xb[u], sp + o0 + 1024u, 16);
        }
    };
    auto stores = [&](const uint4 (&xa)[4], const uint4 (&xb)[4], const uint32_t sg) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t nn = lane_get(ntrue, sg + u);
            uint8_t *dp = lit + lane_get(dstp, sg + u);
            if (o0 < nn) __builtin_memcpy(dp + o0, &xa[u], 16);
            if (o0 + 1024u < nn) __builtin_memcpy(dp + o0 + 1024u, &xb[u], 16);
            if (nn > 2048u) { /* rare: more than 2 KiB from one walker */
                const uint8_t *sp = tbase + lane_get(srcp, sg + u);
                for (uint32_t o = 2048u + o0; o < nn; o += 1024u) {
                    uint4 v;
                    __builtin_memcpy(&v, sp + o, 16);
                    __builtin_memcpy(dp + o, &v, 16);
                }
            }
        }
    };
    loads(va, vb, 0);
#pragma unroll 1
    for (uint32_t sg = 0; sg < 64u; sg += 8u) {
        loads(wa, wb, sg + 4u);
        stores(va, vb, sg);
        if (sg + 8u < 64u) loads(va, vb, sg + 8u);
        stores(wa, wb, sg + 4u);
    }
}

/* ------------------------------------------------------------------------------------------------ K3 */
/* extra bits of a literal-length / match-length code: 0 below 16 / 32, a packed nibble table for the next 9 / 11
 * codes, code - 19 / code - 36 above (the format's tables, RFC 8878 3.1.1.3.2.1.1) */
__device__ inline uint32_t ll_xbits(uint32_t sym)
{
    const uint32_t k = sym - 16u;
    const uint32_t nib = (uint32_t)(0x433221111ull >> (4u * (k & 15u))) & 15u;
    return sym < 16u ? 0u : (sym >= 25u ? sym - 19u : nib);
}
__device__ inline uint32_t ml_xbits(uint32_t sym)
{
    const uint32_t k = sym - 32u;
    const uint32_t nib = (uint32_t)(0x54433221111ull >> (4u * (k & 15u))) & 15u;
    return sym < 32u ? 0u : (sym >= 43u ? sym - 36u : nib);
}

/* ------------------------------------------------------------------------------------------------ K3' */
/* k_zchain4 + k_zmat: the sequence stage split by what is serial in it.
 *
 * Round 2's k_zseq did everything for a sequence on the one lane that owns the block: 321 instructions, and a lone wave per
 * SIMD (LDS capacity: the decoding tables) issues one instruction every ~4.7 cycles -- the stage is as long as its
 * instruction count.  What is serial in the FSE sequence stream is only the chain
 *     three table entries -> bit counts -> position of the state bits -> next states -> next entries;
 * the values (literal length, match length, offset = base + extra bits) and even the repeat-offset history are a
 * function of (bit position, three symbols) per sequence and are computed afterwards, 64 sequences at a time, by a
 * wave per frame (k_zmat) -- the history by a parallel scan over "history transforms".
 *
 * The chain kernel (round 3: k_zchain, a lane per block, 29 blocks' tables in LDS; since round 4 k_zchain4 below -- the lane-
 * per-block kernel is profiles/scripts/r06_removed_variants.patch) stores an 8-byte record per sequence (unread
 * bits before the sequence | the three symbols), takes the state bits out of ONE 8-byte window read from the
 * lane's ring (unaligned ds_read_b64; the rare sequence with more than 57 bits re-reads) and looks the next entries
 * up: one LDS round trip and ~100 instructions per sequence.  The rings (128 bytes per lane, contiguous, first 8 bytes
 * mirrored behind the end) are fed cooperatively like the LZ4 index pass's: in turn J the wave's 64 lanes load one
 * 16-byte piece each for 16 walkers (4 lanes x 16 B = half a cache line per walker) that have room, and store it
 * four turns later; every per-lane condition is evaluated eagerly (NOTEBOOK.md 4.1). */

/* k_zchain4 (round 4): the same chain with a QUAD of lanes per block -- lane 0 the literal-length state, lane 1 the match-
 * length state, lane 2 the offset state, lane 3 none (it helps with the ring and holds its tongue).  With a lane per block a
 * sequence cost 114 wave-instructions, three table entries decoded one after the other on one lane, and 29 lanes of 64 were
 * in use (LDS: 2.5 KB of tables per block; every block of the headline data carries tables of the format's maximum logs).
 * The kernel holds 156 of a CU's 160 KB of LDS while it runs -- nothing else of the pipeline fits beside it -- so its length
 * is the call's: 5 tiles x 3.6 ms of 27 ms.  Here the three lanes decode their entries at once; what crosses lanes is the
 * number of extra bits (to skip) and the state-bit counts in front of a lane's own (DPP quad broadcasts, no LDS); the
 * stream position, the window and the ring bookkeeping are computed by all four lanes alike, and the quad loads its own
 * half-line of input (four 16-byte pieces: no exchange with other lanes).  16 blocks per wave, 43 KB of LDS, three waves per
 * CU; records, checks and verdicts are k_zchain's bit for bit. */
constexpr uint32_t kCqW = 16;
struct ChainQLds {
    uint16_t tab[kCqW][kSeqTblWords];
    uint8_t ring[kCqW * kChStride + 16]; /* + one trash slot for the wave */
};

template <int CTRL>
__device__ inline uint32_t quad_get(uint32_t v) /* lane CTRL / 0x55 of this lane's quad */
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); /* (update_dpp with a zero `old` costs a v_mov per use) */
}

struct ChainQuad {
    int32_t s0, pos, cb, lowh, fillh;
    uint32_t pend;
    uint32_t st;   /* this lane's FSE state */
    uint32_t e;    /* ... and its table entry */
    uint32_t i, nseq, sh;
    uint64_t raw;
    bool rd_ok;
};

typedef uint32_t cq_u32x4 __attribute__((ext_vector_type(4)));
template <int J, bool FEED>
__device__ inline void chainq_turn(uint8_t *ring, const uint16_t *tab, ChainQuad &z, const uint32_t k, const bool mine, const int32_t c,
                                   const uint32_t m1, const uint32_t m2, cq_u32x4 &fd, uint32_t &fa, uint32_t &fm, const uint8_t *gsrc,
                                   const uint32_t vend, const uint32_t myring, const uint32_t trash_lds, uint2 *out, uint2 *trash, bool &bad)
{
    uint32_t wi = 0;
    if (FEED) { /* the quad asks for the next half-line of its own stream */
        const bool want = (z.i < z.nseq) & (z.cb < 64 * (z.lowh + 1)) & (64 * z.lowh > z.s0);
        wi = want ? 1u : 0u;
        z.lowh -= (int32_t)wi;
    }
    uint2 rec;
    bool go;
    {
        const bool on = z.i < z.nseq;
        go = on & z.rd_ok;
        const uint32_t e = z.e;
        const uint32_t x = e >> 11, v = e & 1023u;
        const uint32_t n = mine ? (uint32_t)(__builtin_clz(v) + c) : 0u;
        const uint32_t x0 = quad_get<0x00>(x), x1 = quad_get<0x55>(x), x2 = quad_get<0xAA>(x);
        const uint32_t n0 = quad_get<0x00>(n), n1 = quad_get<0x55>(n), n2 = quad_get<0xAA>(n);
        const uint32_t X = x0 + x1 + x2, N = n0 + n1 + n2;
        const uint32_t pre = (n0 & m1) + (n1 & m2); /* state bits in front of this lane's: LL, then ML, then OF */
        const uint64_t win = z.raw << z.sh;
        uint32_t W = (uint32_t)((win << X) >> 32);
        const bool ovf = go & (X + N > 57u);
        if (ovf) { /* more bits than one window holds: the state bits come from a second read */
            const int32_t p2 = z.pos - (int32_t)X;
            const int32_t cb2 = z.s0 + ((p2 - 1) >> 3);
            uint64_t r2;
            __builtin_memcpy(&r2, ring + myring + ((uint32_t)(cb2 - 7) & (kChRing - 1u)), 8);
            W = (uint32_t)((r2 << (7u - ((uint32_t)(p2 - 1) & 7u))) >> 32);
        }
        const uint32_t b = __builtin_amdgcn_ubfe(W, 32u - pre - n, n);
        const uint32_t ns = mine ? (v << n) - (1u << (c + 31)) + b : 0u;
        const int32_t npos = z.pos - (int32_t)(X + N);
        /* the record, as k_zchain writes it: unread bits | LL state << 20, OF state | ML state << 8 (lane 0's is stored) */
        const uint32_t sm = quad_get<0x55>(z.st), so = quad_get<0xAA>(z.st);
        rec = make_uint2((uint32_t)z.pos | (z.st << 20), so | (sm << 8));
        z.st = go ? ns : z.st;
        z.pos = go ? npos : z.pos;
        const bool neg = go & (npos < 0); /* read past the start of the stream */
        bad = bad | neg;
        z.nseq = neg ? 0u : z.nseq;
    }
    z.cb = z.s0 + ((z.pos - 1) >> 3);
    z.sh = 7u - ((uint32_t)(z.pos - 1) & 7u);
    z.rd_ok = (z.cb - 16 >= 64 * z.fillh) | (64 * z.fillh <= z.s0);
    __builtin_memcpy(&z.raw, ring + myring + ((uint32_t)(z.cb - 7) & (kChRing - 1u)), 8);
    z.e = tab[z.st];
    *((go & (k == 0u)) ? out + z.i : trash) = rec;
    z.i += go ? 1u : 0u;
    if (!FEED) return;
    /* The piece slot J's load brought an iteration ago.  The load is inline assembly and its wait is written by hand (as in
     * lz4_index.hip): left to the compiler, the four slots were copied between register sets at the loop's latch, behind
     * s_waitcnt vmcnt(9) / (6) / (3) / (0) -- every iteration of eight sequences drained the load issued a turn earlier, a
     * trip to memory on the chain of a wave that has its SIMD to itself.  An iteration issues 8 record stores and 4 loads;
     * between slot J's load and its commit lie exactly 11 younger operations. */
    asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    *reinterpret_cast<cq_u32x4 *>(ring + fa) = fd;
    *reinterpret_cast<uint2 *>(ring + fm) = make_uint2(fd[0], fd[1]);
    {
        const uint32_t got = (z.pend >> J) & 1u;
        z.fillh -= (int32_t)got;
        z.pend = (z.pend & ~(1u << J)) | (wi << J);
    }
    {
        const bool p = wi != 0u;
        const uint32_t o = ((uint32_t)z.lowh << 6) + k * 16u;
        fa = p ? myring + (o & (kChRing - 1u)) : trash_lds;
        fm = (p & ((o & (kChRing - 1u)) == 0u)) ? myring + kChRing : trash_lds;
        const uint8_t *g = gsrc + ((p & (o < vend)) ? o : 0u); /* always one load per turn (k_zchain) */
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(fd) : "v"(g));
    }
}

__global__ void __launch_bounds__(64) k_zchain4(ZPipe P)
{
    __shared__ __attribute__((aligned(16))) ChainQLds L;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nitems = uni(P.counters[3]);
    /* grid-stride: the grid is sized for the chip (768 waves), not for frames x blocks-a-frame-may-have descriptor slots */
    for (uint32_t i0 = blockIdx.x * kCqW; i0 < nitems; i0 += gridDim.x * kCqW) {
    if (i0 != blockIdx.x * kCqW) { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }
    for (uint32_t j = 0; j < kCqW && i0 + j < nitems; j++) {
        const uint32_t it = uni(P.sitems[i0 + j]);
        const uint32_t fj = it / P.nbmax;
        const ZBlk *d = P.blks + it;
        const uint32_t slots = uni(d->slots), logs = uni(d->logs);
#pragma unroll
        for (int kind = 0; kind < 3; kind++) {
            const uint32_t slot = (slots >> (8 * kind)) & 255u, lg = (logs >> (8 * kind)) & 255u;
            const uint32_t goff = kind == 0 ? 0u : (kind == 1 ? 1024u : 512u);
            const uint32_t *g = (slot == kPredefSlot ? P.predef : P.seqt + ((uint64_t)fj * P.nbmax + slot) * kSeqTblWords) + goff;
            for (uint32_t q = lane; q < (1u << lg); q += 64u) { /* 25-bit workspace entry -> 16 bits: extra-bit count (5) | v (10) */
                const uint32_t e = g[q], nb = (e >> 10) & 15u;
                L.tab[j][goff + q] = (uint16_t)((((e >> 20) & 31u) << 11) | (1u << (lg - nb)) | ((e & 1023u) >> nb));
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    const uint32_t q = lane >> 2, k = lane & 3u;
    bool act = i0 + q < nitems;
    const uint32_t it = act ? P.sitems[i0 + q] : P.sitems[i0];
    const uint32_t f = it / P.nbmax;
    const ZBlk *d = P.blks + it;
    if (act && (P.frames[f].flags & (F_BAD | F_IRREG))) act = false;
    const uint64_t fo = P.src_off[P.first + f];
    const uint64_t aoff = fo & ~(uint64_t)63;
    const uint32_t delta = (uint32_t)(fo & 63u);
    const uint32_t vend = delta + P.src_size[P.first + f];
    const uint32_t logs = d->logs;
    const uint32_t sq_off = d->sq_off, sq_len = d->sq_len;
    const uint8_t *gmine = P.src_base + aoff;
    const uint32_t myring = q * kChStride;
    const uint32_t trash_lds = kCqW * kChStride;
    /* kinds in the order their state bits lie in the stream: LL (lane 0), ML (lane 1), OF (lane 2) */
    const uint32_t lgl = logs & 255u, lgo = (logs >> 8) & 255u, lgm = (logs >> 16) & 255u;
    const bool mine = k < 3u;
    const uint32_t lgk = k == 0u ? lgl : (k == 1u ? lgm : lgo);
    const int32_t c = (int32_t)lgk - 31;
    const uint32_t m1 = k >= 1u ? ~0u : 0u, m2 = k >= 2u ? ~0u : 0u;
    const uint16_t *tab = L.tab[q] + (k == 0u ? 0u : (k == 1u ? 512u : (k == 2u ? 1024u : 0u)));
    ChainQuad z;
    z.s0 = (int32_t)(delta + sq_off);
    z.pos = 0; z.cb = z.s0; z.lowh = 0; z.fillh = 0; z.pend = 0; z.st = 0; z.e = 0; z.i = 0; z.nseq = 0; z.sh = 0; z.raw = 0;
    bool bad = false;
    bool opened = act && sq_len >= 1u;
    uint32_t last = 0;
    if (opened) last = gmine[delta + sq_off + sq_len - 1u];
    if (last == 0u) opened = false;
    if (act && !opened) bad = true;
    if (opened) {
        /* the stream's top two half-lines: each lane of the quad its piece */
        const int32_t s1 = z.s0 + (int32_t)sq_len;
        const int32_t ht = (s1 - 1) >> 6;
        const int32_t hl = ht >= 1 ? ht - 1 : 0;
        for (int32_t h = hl; h <= ht; h++) {
            const uint32_t o = (uint32_t)h * 64u + k * 16u;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (o < vend) v = *reinterpret_cast<const uint4 *>(gmine + o);
            *reinterpret_cast<uint4 *>(L.ring + myring + (o & (kChRing - 1u))) = v;
            if ((o & (kChRing - 1u)) == 0u) *reinterpret_cast<uint2 *>(L.ring + myring + kChRing) = make_uint2(v.x, v.y);
        }
        z.lowh = z.fillh = hl;
        z.nseq = d->nseq - 1u; /* the turns take every sequence that is followed by state bits; the block's last one after the loop */
        z.pos = (int32_t)(sq_len - 1u) * 8 + (31 - __builtin_clz(last));
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    if (opened) {
        /* initial states: LL, OF, ML from the top of the stream (<= 26 bits) */
        const int32_t cb = z.s0 + ((z.pos - 1) >> 3);
        uint64_t r;
        __builtin_memcpy(&r, L.ring + myring + ((uint32_t)(cb - 7) & (kChRing - 1u)), 8);
        const uint32_t W = (uint32_t)((r << (7u - ((uint32_t)(z.pos - 1) & 7u))) >> 32);
        const uint32_t sl = __builtin_amdgcn_ubfe(W, 32u - lgl, lgl);
        const uint32_t so = __builtin_amdgcn_ubfe(W, 32u - lgl - lgo, lgo);
        const uint32_t sm = __builtin_amdgcn_ubfe(W, 32u - lgl - lgo - lgm, lgm);
        z.st = k == 0u ? sl : (k == 1u ? sm : (k == 2u ? so : 0u));
        z.pos -= (int32_t)(lgl + lgo + lgm);
        if (z.pos < 0) { bad = true; z.nseq = 0; z.pos = 0; opened = false; }
    }
    __builtin_amdgcn_wave_barrier();
    z.cb = z.s0 + ((z.pos - 1) >> 3);
    z.sh = 7u - ((uint32_t)(z.pos - 1) & 7u);
    z.rd_ok = true; /* the top of the stream is in the ring */
    __builtin_memcpy(&z.raw, L.ring + myring + ((uint32_t)(z.cb - 7) & (kChRing - 1u)), 8);
    z.e = mine ? tab[z.st] : 0u;

    uint2 *out = P.chain + (opened ? d->seq_base : 0u);
    uint2 *trash = P.chain + P.seqcap + lane;
    cq_u32x4 fd0 = {0, 0, 0, 0}, fd1 = fd0, fd2 = fd0, fd3 = fd0;
    uint32_t fa0 = trash_lds, fa1 = trash_lds, fa2 = trash_lds, fa3 = trash_lds, fm0 = trash_lds, fm1 = trash_lds, fm2 = trash_lds, fm3 = trash_lds;
#define CQ_TURN(J, FEED, FD, FA, FM) chainq_turn<J, FEED>(L.ring, tab, z, k, mine, c, m1, m2, FD, FA, FM, gmine, vend, myring, trash_lds, out, trash, bad);
    if (wave_any(z.i < z.nseq)) {
        do { /* (do-while: with `while` the register allocator copies the slots at the top of the loop, lz4_index.hip) */
            CQ_TURN(0, true, fd0, fa0, fm0) CQ_TURN(0, false, fd0, fa0, fm0)
            CQ_TURN(1, true, fd1, fa1, fm1) CQ_TURN(1, false, fd1, fa1, fm1)
            CQ_TURN(2, true, fd2, fa2, fm2) CQ_TURN(2, false, fd2, fa2, fm2)
            CQ_TURN(3, true, fd3, fa3, fm3) CQ_TURN(3, false, fd3, fa3, fm3)
        } while (wave_any(z.i < z.nseq));
    }
    /* the compiler does not know of the assembly loads: the drain names the slots, or their registers -- dead to the
     * compiler once the loop is left -- are handed to something else while a load is still on its way into them (seen: an
     * address computed into two of them in front of the drain, overwritten by the late load: a store into the wild) */
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(fd0), "+v"(fd1), "+v"(fd2), "+v"(fd3) : : "memory");
#undef CQ_TURN
    if (opened && !bad) { /* the last sequence: its extra bits, no state bits */
        const uint32_t x = z.e >> 11;
        const uint32_t X = quad_get<0x00>(x) + quad_get<0x55>(x) + quad_get<0xAA>(x);
        const uint32_t sm = quad_get<0x55>(z.st), so = quad_get<0xAA>(z.st);
        if (k == 0u) out[z.i] = make_uint2((uint32_t)z.pos | (z.st << 20), so | (sm << 8));
        z.pos -= (int32_t)X;
    }
    if (opened && z.pos != 0) bad = true; /* the bitstream must be consumed exactly */
    if (bad && k == 0u) atomicOr(&P.frames[f].flags, F_BAD);
    }
}

/* K3'': values and repeat offsets, one wave per frame, 64 sequences at a time.  A sequence changes the offset history
 * (r0, r1, r2) by one of five "transforms" whose outputs are each either a constant (a fresh offset) or one of the
 * inputs minus a small count, floored at 1 (the format's `rep0 - 1`, never 0); such transforms compose into the same
 * shape, so the history in front of every sequence is an exclusive scan over the batch applied to the carry. */
/* one word per output slot: value << 2 | tag; tag 0..2 = input slot (value = count subtracted), 3 = constant.  Offsets
 * of 2^30 - 1 and more are kept as 2^30 - 1: the pipeline only decodes frames of at most 254 blocks (31.75 MiB), so both
 * are farther back than any output position and k_zexec rejects them alike. */
struct RepT { uint32_t s0, s1, s2; };
constexpr uint32_t kRepMax = 0x3FFFFFFFu;

__device__ inline uint32_t rep_apply(uint32_t slot, uint32_t h0, uint32_t h1, uint32_t h2)
{
    const uint32_t tag = slot & 3u, v = slot >> 2;
    const uint32_t h = tag == 0u ? h0 : (tag == 1u ? h1 : h2);
    const uint32_t dec = h > v ? h - v : 1u;
    return tag == 3u ? v : dec;
}
/* slot b of the later transform applied to the outputs A of the earlier one */
__device__ inline uint32_t rep_compose1(const RepT &A, uint32_t b)
{
    const uint32_t tb = b & 3u, vb = b >> 2;
    const uint32_t a = tb == 0u ? A.s0 : (tb == 1u ? A.s1 : A.s2);
    const uint32_t ta = a & 3u, va = a >> 2;
    const uint32_t cdec = va > vb ? va - vb : 1u;
    const uint32_t r = ((ta == 3u ? cdec : va + vb) << 2) | ta;
    return tb == 3u ? b : r;
}

__global__ void __launch_bounds__(64) k_zmat(ZPipe P)
{
    __shared__ uint32_t s_llb[36], s_mlb[53];
    __shared__ uint2 s_op[64 * 9];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t f = blockIdx.x;
    if (lane < 36u) s_llb[lane] = kLLBase[lane];
    if (lane < 53u) s_mlb[lane] = kMLBase[lane];
    __builtin_amdgcn_wave_barrier();
    if (uni(P.frames[f].flags) & (F_BAD | F_IRREG)) return;
    const uint32_t nblk = uni(P.frames[f].nblk);
    const uint8_t *src = P.src_base + uni64(P.src_off[P.first + f]);
    uint32_t h0 = 1, h1 = 4, h2 = 8; /* wave-uniform */
    for (uint32_t k = 0; k < nblk; k++) {
        const ZBlk *d = P.blks + (uint64_t)f * P.nbmax + k;
        const uint32_t n = uni(d->nseq);
        if (uni(d->type) != 2u || n == 0u) continue;
        const uint2 *cr = P.chain + uni(d->seq_base);
        uint2 *out = P.seqs + uni(d->seq_base);
        const uint8_t *sq = src + uni(d->sq_off);
        const uint32_t slots = uni(d->slots);
        const uint32_t *tl = (slots & 255u) == kPredefSlot ? P.predef : P.seqt + ((uint64_t)f * P.nbmax + (slots & 255u)) * kSeqTblWords;
        const uint32_t *to = (((slots >> 8) & 255u) == kPredefSlot ? P.predef : P.seqt + ((uint64_t)f * P.nbmax + ((slots >> 8) & 255u)) * kSeqTblWords) + 1024u;
        const uint32_t *tm = (((slots >> 16) & 255u) == kPredefSlot ? P.predef : P.seqt + ((uint64_t)f * P.nbmax + ((slots >> 16) & 255u)) * kSeqTblWords) + 512u;
        /* 512 sequences per round.  Memory is touched 64 consecutive sequences at a time (lane l: base + 64 e + l); the
         * history needs consecutive sequences per lane (lane l: base + 8 l + e): what a sequence does to the history
         * (kind, offset value) crosses over through LDS, the resulting offset comes back the same way.  The lane folds
         * its eight into one transform, the wave scans the 64 transforms, the lane walks its eight with the real history. */
        constexpr uint32_t kPer = 8;
        for (uint32_t base = 0; base < n; base += 64u * kPer) {
            uint32_t ll[kPer], ml[kPer];
#pragma unroll
            for (uint32_t e = 0; e < kPer; e++) {
                const uint32_t q = 64u * e + lane;
                const bool on = base + q < n;
                uint2 c = make_uint2(0, 0);
                if (on) c = cr[base + q];
                /* states -> symbols, from the block's tables in the workspace (symbol at bits 14..19) */
                const uint32_t pos = c.x & 0xFFFFFu;
                uint32_t lsym = 0, osym = 0, msym = 0;
                if (on) {
                    lsym = (tl[c.x >> 20] >> 14) & 63u;
                    osym = (to[c.y & 255u] >> 14) & 31u;
                    msym = (tm[c.y >> 8] >> 14) & 63u;
                }
                const uint32_t llb = ll_xbits(lsym), mlb = ml_xbits(msym);
                const uint32_t lo = pos - osym, lo2 = lo - llb - mlb; /* valid frames: lo2 >= 0 */
                uint32_t extra = 0, llv = 0, mlv = 0;
                if (on) {
                    const uint64_t a = ld64v(sq + (lo >> 3)), b = ld64v(sq + (lo2 >> 3));
                    extra = (uint32_t)(a >> (lo & 7u)) & (uint32_t)((1ull << osym) - 1ull);
                    const uint32_t t = (uint32_t)(b >> (lo2 & 7u));
                    llv = t & ((1u << llb) - 1u);
                    mlv = (t >> llb) & ((1u << mlb) - 1u);
                }
                ll[e] = s_llb[lsym < 36u ? lsym : 0u] + llv;
                ml[e] = s_mlb[msym < 53u ? msym : 0u] + mlv;
                const bool fresh = osym > 1u;
                const uint32_t idx = (osym == 1u ? 1u + extra : 0u) + (lsym == 0u ? 1u : 0u); /* repeat-offset index 0..3 */
                /* kind: 0..3 repeat index, 4 fresh, 5 no sequence */
                s_op[(q >> 3) * 9u + (q & 7u)] = make_uint2(((1u << osym) - 3u) + extra, !on ? 5u : (fresh ? 4u : idx));
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            uint32_t vv[kPer], kd[kPer];
            RepT T;
            T.s0 = 0u; T.s1 = 1u; T.s2 = 2u;
#pragma unroll
            for (uint32_t e = 0; e < kPer; e++) { /* this sequence after the lane's earlier ones */
                const uint2 o = s_op[lane * 9u + e];
                const uint32_t v = o.x, k = o.y;
                vv[e] = v; kd[e] = k;
                const uint32_t vs = v < kRepMax ? v : kRepMax;
                const uint32_t t0 = T.s0, t1 = T.s1, t2 = T.s2;
                const uint32_t tg = t0 & 3u, tv = t0 >> 2;
                const uint32_t dec = tg == 3u ? ((tv > 1u ? tv - 1u : 1u) << 2) | 3u : t0 + 4u; /* max(1, . - 1) */
                T.s0 = k == 4u ? ((vs << 2) | 3u) : (k == 1u ? t1 : (k == 2u ? t2 : (k == 3u ? dec : t0)));
                T.s1 = (k >= 1u && k <= 4u) ? t0 : t1;
                T.s2 = (k >= 2u && k <= 4u) ? t1 : t2;
            }
            /* inclusive scan over the lanes */
#pragma unroll
            for (int st = 1; st < 64; st <<= 1) {
                RepT A;
                A.s0 = (uint32_t)__shfl_up((int)T.s0, st, 64);
                A.s1 = (uint32_t)__shfl_up((int)T.s1, st, 64);
                A.s2 = (uint32_t)__shfl_up((int)T.s2, st, 64);
                const uint32_t c0 = rep_compose1(A, T.s0), c1 = rep_compose1(A, T.s1), c2 = rep_compose1(A, T.s2);
                const bool take = lane >= (uint32_t)st;
                T.s0 = take ? c0 : T.s0; T.s1 = take ? c1 : T.s1; T.s2 = take ? c2 : T.s2;
            }
            /* the history in front of this lane's first sequence: the lanes before it applied to the carry */
            uint32_t e0 = (uint32_t)__shfl_up((int)T.s0, 1, 64), e1 = (uint32_t)__shfl_up((int)T.s1, 1, 64), e2 = (uint32_t)__shfl_up((int)T.s2, 1, 64);
            e0 = lane == 0u ? 0u : e0; e1 = lane == 0u ? 1u : e1; e2 = lane == 0u ? 2u : e2;
            uint32_t r0 = rep_apply(e0, h0, h1, h2), r1 = rep_apply(e1, h0, h1, h2), r2 = rep_apply(e2, h0, h1, h2);
#pragma unroll
            for (uint32_t e = 0; e < kPer; e++) {
                const uint32_t k = kd[e];
                const uint32_t dm = r0 > 1u ? r0 - 1u : 1u;
                const uint32_t off = k == 4u ? vv[e] : (k == 1u ? r1 : (k == 2u ? r2 : (k == 3u ? dm : r0)));
                /* the history itself is kept like the transforms keep it (offsets beyond 2^30 - 1 as 2^30 - 1) */
                const uint32_t offh = k == 4u ? (vv[e] < kRepMax ? vv[e] : kRepMax) : off;
                const uint32_t q0 = r0, q1 = r1;
                r0 = (k >= 1u && k <= 4u) ? offh : r0;
                r1 = (k >= 1u && k <= 4u) ? q0 : r1;
                r2 = (k >= 2u && k <= 4u) ? q1 : r2;
                s_op[lane * 9u + e].x = off;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (uint32_t e = 0; e < kPer; e++) {
                const uint32_t q = 64u * e + lane;
                const uint32_t off = s_op[(q >> 3) * 9u + (q & 7u)].x;
                const uint32_t o29 = off < 0x1FFFFFFFu ? off : 0x1FFFFFFFu;
                if (base + q < n) out[base + q] = make_uint2(o29 | (ll[e] << 29), (ll[e] >> 3) | (ml[e] << 14));
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            /* carry: the whole round applied to the history */
            const uint32_t n0 = rep_apply(lane_get(T.s0, 63), h0, h1, h2), n1 = rep_apply(lane_get(T.s1, 63), h0, h1, h2),
                           n2 = rep_apply(lane_get(T.s2, 63), h0, h1, h2);
            h0 = uni(n0); h1 = uni(n1); h2 = uni(n2);
        }
    }
}

/* ------------------------------------------------------------------------------------------------ K4 */
namespace {

/* output bytes per batch: the largest T with R - T >= T + 1023 at R = 4096 (lz4_dec2.hip); a batch's literals have to
 * fit the input ring next to the chunk being staged */
constexpr uint32_t kZR = ZR, kZT = (kZR - 1024u) / 2u, kZLitMax = kInRing - kInChunk - 16u; /* (the copy engine keeps positions in 11 bits: T < 2048 whatever the ring) */

struct ExecLds {
    uint8_t ring[kZR + 16];   /* + the copy engine's 16-byte tail (lz4_copy.h) */
    uint8_t in[kInRing + 16];
    uint32_t meta[64];
    uint32_t bm[CopyLds<kZR, kZT>::kWords];
};

/* execute one compressed block's sequences; false on malformed input */
__device__ bool exec_block(ExecLds &L, Wave<kZR> &w, const ZPipe &P, const ZBlk *d, const uint8_t *src, uint32_t f,
                           uint32_t cap, uint32_t lane, Stats &st)
{
    const uint32_t regen = uni(d->regen), nseq = uni(d->nseq);
    const int lit_mode = (int)uni(d->lit_mode);
    const uint32_t lit_src = uni(d->lit_src);
    const uint32_t rle_byte = lit_src;
    uint32_t lit_pos = 0, lvp = 0;
    if (lit_mode == 0) lvp = stream_open(w, src + lit_src, regen);
    else if (lit_mode == 2) {
        if (uni(d->segd)) { /* the walkers' pieces, where k_zhufw left them */
            const uint32_t it = (uint32_t)(d - P.blks), kblk = it - f * P.nbmax;
            const uint8_t *tbase = P.htmp + (uint64_t)f * P.htmp_stride + ((2u * (uint64_t)lit_src + 127u) & ~(uint64_t)127) + (uint64_t)kblk * kHwBlockSlack;
            lvp = stream_open_seg(w, tbase, P.hsegs[(uint64_t)it * 64u + lane], regen);
        } else lvp = stream_open(w, P.lits + (uint64_t)f * P.litcap + lit_src, regen);
    }
    const bool streamed = lit_mode != 1;

    const uint2 *seqs = P.seqs + uni(d->seq_base);
    uint32_t q_ll = 0, q_ml = 0, q_off = 0;
    uint32_t qn = 0, loaded = 0;
    uint2 nx = make_uint2(0, 0); /* records loaded .. loaded+63, one per lane, in flight */
    if (lane < nseq) nx = seqs[lane];
    for (;;) {
        /* ---- top the queue up from the prefetched records ---- */
        if (qn < 64u && loaded < nseq) {
            uint32_t take = 64u - qn;
            if (take > nseq - loaded) take = nseq - loaded;
            const int from = (int)((lane - qn) & 63u);
            const uint32_t rx = (uint32_t)__shfl((int)nx.x, from, 64), ry = (uint32_t)__shfl((int)nx.y, from, 64);
            if (lane >= qn && lane < qn + take) { q_ll = (rx >> 29) | ((ry & 0x3FFFu) << 3); q_ml = ry >> 14; q_off = rx & 0x1FFFFFFFu; }
            qn += take;
            loaded += take;
            nx = make_uint2(0, 0);
            if (loaded + lane < nseq) nx = seqs[loaded + lane];
        }
        if (qn == 0u) break;

        /* ---- head prefix of the queue that the batch engine can take ---- */
        const bool inq = lane < qn;
        const uint32_t outlen = inq ? q_ll + q_ml : 0u;
        const uint32_t oend = scan64_incl(outlen);
        const uint32_t ostart = oend - outlen;
        const uint32_t litend = scan64_incl(inq ? q_ll : 0u);
        const uint32_t mabs = w.op + ostart + q_ll;
        const bool isfar = inq && q_off >= kZR - kZT; /* in the ring for the whole batch, or flushed before it (lz4_copy.h) */
        const bool ok = inq && streamed && (q_ml <= q_off || (q_off != 0u && q_ml <= 64u)) /* short self-overlap: a dependent match, lz4_copy.h */ && q_off <= mabs && !(isfar && q_ml > 32u) &&
                        litend <= regen - lit_pos && litend <= kZLitMax && oend <= kZT && (uint64_t)w.op + oend <= cap;
        const unsigned long long badmask = wave_ballot(!ok);
        const uint32_t nb = badmask ? ctz64(badmask) : 64u;
        if (nb > 0u) {
            const uint32_t T = lane_get(oend, nb - 1u);
            const uint32_t lits = lane_get(litend, nb - 1u);
            while (w.in_hi < w.vend && w.in_hi < lvp + lits + 8u) w.refill();
            /* the copy engine of the LZ4 decoders (lz4_copy.h): literals and independent matches one lane per
             * sequence, dependent matches byte per lane; the rings' first 16 bytes are mirrored behind them */
            asm volatile("" ::: "memory");
            if (lane < 2u) *reinterpret_cast<uint2 *>(L.in + kInRing + lane * 8u) = *reinterpret_cast<const uint2 *>(L.in + lane * 8u);
            else if (lane < 4u) *reinterpret_cast<uint2 *>(L.ring + kZR + (lane - 2u) * 8u) = *reinterpret_cast<const uint2 *>(L.ring + (lane - 2u) * 8u);
            w.flush();
            uint4 xfa = make_uint4(0, 0, 0, 0), xfb = xfa;
            if (lane < nb && isfar) {
                const uint8_t *g = w.dst + (mabs - q_off);
                __builtin_memcpy(&xfa, g, 16);
                __builtin_memcpy(&xfb, g + 16, 16);
            }
            const CopyLds<kZR, kZT> SL = {L.ring, L.in, L.meta, L.bm};
            seq_copy<kZR, kZT>(w, SL, nb, ostart, q_ll, q_ml, q_off, lvp + (litend - q_ll), T, isfar, xfa, xfb, st);
            w.flush();
            lvp += lits;
            lit_pos += lits;
            q_ll = __shfl(q_ll, (int)((lane + nb) & 63u), 64);
            q_ml = __shfl(q_ml, (int)((lane + nb) & 63u), 64);
            q_off = __shfl(q_off, (int)((lane + nb) & 63u), 64);
            qn -= nb;
        } else {
            /* not batchable (overlapping or very long match, RLE literals, or malformed): alone, every check */
            const uint32_t llen = lane_get(q_ll, 0), mlen = lane_get(q_ml, 0), offset = lane_get(q_off, 0);
            if (llen > regen - lit_pos) return false;
            if ((uint64_t)llen + mlen > (uint64_t)(cap - w.op)) return false;
            if (streamed) lvp = wave_copy_literals(w, lvp, llen);
            else wave_fill(w, rle_byte, llen);
            lit_pos += llen;
            if (offset > w.op) return false;
            wave_copy_match(w, offset, mlen);
            w.flush();
            q_ll = __shfl(q_ll, (int)((lane + 1u) & 63u), 64);
            q_ml = __shfl(q_ml, (int)((lane + 1u) & 63u), 64);
            q_off = __shfl(q_off, (int)((lane + 1u) & 63u), 64);
            qn -= 1u;
        }
    }
    const uint32_t rest = regen - lit_pos;
    if (rest > cap - w.op) return false;
    if (streamed) wave_copy_literals(w, lvp, rest);
    else wave_fill(w, rle_byte, rest);
    w.flush();
    return true;
}

} // namespace

__global__ void __launch_bounds__(64, kZexecOcc) k_zexec(ZPipe P)
{
    __shared__ __attribute__((aligned(16))) ExecLds L;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t f = blockIdx.x;
    const uint32_t flags = uni(P.frames[f].flags);
    if (flags & F_IRREG) return; /* the fused decoder writes this block and its status */
    if (P.done != nullptr && uni(P.done[f]) != 0u) return; /* decoded, and its status written, by the few-frames path below */
    const uint64_t blk = P.first + f;
    if (flags & F_BAD) {
        if (lane == 0) P.status[blk] = CRYO_ST_CORRUPT;
        return;
    }
    const uint8_t *src = P.src_base + uni64(P.src_off[blk]);
    const uint32_t nblk = uni(P.frames[f].nblk);
    const uint32_t B = P.B;
    Stats st = {};
    Wave<kZR> w;
    w.ring = L.ring;
    w.in = L.in;
    w.lane = lane;
    w.dst = P.dst_base + uni64(blk * P.dst_stride);
    w.dst_aligned = (reinterpret_cast<uintptr_t>(w.dst) & 15u) == 0;
    w.op = 0;
    w.flushed = 0;
    w.delta = 0; w.abase = src; w.vend = 0; w.in_hi = 0; w.pre = make_uint2(0, 0); w.pre2 = make_uint2(0, 0); w.pre3 = make_uint2(0, 0); w.nstale = 0;

    bool bad = false;
    for (uint32_t k = 0; k < nblk; k++) {
        const ZBlk *d = P.blks + (uint64_t)f * P.nbmax + k;
        const uint32_t type = uni(d->type), bsize = uni(d->bsize), so = uni(d->src_off);
        if (type == 1u) {
            if (bsize > B - w.op) { bad = true; break; }
            wave_fill(w, uni(src[so]), bsize);
            w.flush();
        } else if (type == 0u) {
            if (bsize > B - w.op) { bad = true; break; }
            const uint32_t vp = stream_open(w, src + so, bsize);
            wave_copy_literals(w, vp, bsize);
            w.flush();
        } else {
            if (!exec_block(L, w, P, d, src, f, B, lane, st)) { bad = true; break; }
        }
    }
    if (!bad && (flags & F_FCS)) {
        const uint64_t fcs = (uint64_t)uni(P.frames[f].fcs_lo) | ((uint64_t)uni(P.frames[f].fcs_hi) << 32);
        if ((uint64_t)w.op != fcs) bad = true;
    }
    if (!bad && (flags & F_CK)) {
        w.flush();
        w.flush_tail();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        uint32_t want;
        __builtin_memcpy(&want, src + uni(P.frames[f].ck_off), 4);
        if ((uint32_t)xxh64_dev(w.dst, w.op) != uni(want)) bad = true;
    }
    if (!bad && w.op != B) bad = true;
    if (!bad) { w.flush(); w.flush_tail(); }
    if (lane == 0) P.status[blk] = bad ? CRYO_ST_CORRUPT : CRYO_ST_OK;
}

/* ------------------------------------------------------------------------------------------- few frames per call
 * k_zexec executes a frame's sequences on ONE wave: 800 batches for a 1 MiB frame, 3.8 of the 4.4 ms of that call -- and one
 * frame per call is the reference's own read path (cache.c:178, default codec zstd, CRYO_BLCKSZ 1 MiB).  Once k_zmat has made
 * the sequences explicit, executing them is the problem lz4_lat.hip solves for LZ4 blocks: output positions by a prefix sum,
 * literal bytes placed and every match byte pointed at its source, pointer jumping, gather (lat_copy.h).  For calls of up
 * to 64 frames and 64 MiB:
 *   k_zlat_count  per frame: is it one this path takes (no checksum to verify, no RLE literals, it says B bytes), the
 *                 place of every zstd block's sequences in the frame's flat list
 *   k_zlat_build  per zstd block: its records -> ll / ml / offset / where the literals lie (input: raw literals, raw and
 *                 RLE blocks; bit 31: the frame's pool of Huffman literals), by a scan of the literal lengths; the block's
 *                 last literals as a sequence without a match; a raw block is one literal run, an RLE block one literal
 *                 and a match at offset 1
 *   k_zlat_sum / k_lat_scan / k_zlat_place  output positions; offsets inside the output so far; the frame decodes to B bytes
 *   k_lat_fill / k_lat_jump / k_lat_gather  the bytes
 * Whatever fails a check here is not marked done and k_zexec decodes it as before: verdicts and bytes are its. */
namespace {

__global__ void __launch_bounds__(64) k_zlat_count(ZPipe P, LatArgs A, uint32_t *zstart)
{
    const uint32_t f = blockIdx.x;
    if (threadIdx.x != 0u) return;
    const ZFrame fr = P.frames[f];
    bool ok = (fr.flags & (F_IRREG | F_BAD | F_CK)) == 0u;
    if ((fr.flags & F_FCS) && !(fr.fcs_hi == 0u && fr.fcs_lo == P.B)) ok = false;
    uint32_t total = 0;
    for (uint32_t k = 0; k < fr.nblk; k++) {
        const ZBlk *d = P.blks + (uint64_t)f * P.nbmax + k;
        zstart[f * P.nbmax + k] = total;
        if (d->type == 2u) {
            if (d->lit_mode == 1u && d->regen != 0u) ok = false; /* RLE literals: no stream to point at */
            total += d->nseq + 1u;
        } else total += 1u;
    }
    if (total == 0u || total > A.nmax) ok = false;
    /* sequences of 64 bytes and more on average are runs (the zero gap of narrow rows: RLE blocks, one long match): a memset
     * for k_zexec, a million-deep chain for pointer jumping (16 x 1 MiB `narrow`: 0.55 ms there, 0.98 here) */
    if ((uint64_t)total * 64u < P.B) ok = false;
    A.nseq[f] = total;
    A.ok[f] = ok ? 1u : 0u;
    A.done[f] = 0u;
}

__global__ void __launch_bounds__(256) k_zlat_build(ZPipe P, LatArgs A, const uint32_t *zstart)
{
    __shared__ uint32_t s_sum[4];
    const uint32_t f = blockIdx.y, k = blockIdx.x;
    if (A.ok[f] == 0u || k >= P.frames[f].nblk) return;
    const ZBlk *d = P.blks + (uint64_t)f * P.nbmax + k;
    const uint64_t q0 = (uint64_t)f * A.nmax + zstart[f * P.nbmax + k];
    const uint32_t type = d->type;
    if (type != 2u) {
        if (threadIdx.x == 0u) {
            const uint32_t bs = d->bsize;
            if (type == 0u) { A.ll[q0] = bs; A.ml[q0] = 0u; A.off[q0] = 0u; A.lpos[q0] = d->src_off; }
            else { A.ll[q0] = bs ? 1u : 0u; A.ml[q0] = bs ? bs - 1u : 0u; A.off[q0] = 1u; A.lpos[q0] = d->src_off; } /* RLE: the byte, then a match at distance 1 */
        }
        return;
    }
    const uint32_t nseq = d->nseq, regen = d->regen;
    const uint32_t lbase = d->lit_mode == 2u ? (0x80000000u | d->lit_src) : d->lit_src; /* the pool / the frame's input */
    const uint2 *seqs = P.seqs + d->seq_base;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t carry = 0; /* literal bytes before this chunk of 256 sequences */
    bool bad = false;
    for (uint32_t i0 = 0; i0 < nseq; i0 += 256u) {
        const uint32_t i = i0 + threadIdx.x;
        uint32_t ll = 0, ml = 0, off = 0;
        if (i < nseq) {
            const uint2 r = seqs[i];
            ll = (r.x >> 29) | ((r.y & 0x3FFFu) << 3);
            ml = r.y >> 14;
            off = r.x & 0x1FFFFFFFu;
        }
        const uint32_t incl = scan64_incl(ll);
        __syncthreads(); /* (s_sum of the chunk before has been read) */
        if (lane == 63u) s_sum[wv] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < wv; w++) before += s_sum[w];
        if (i < nseq) {
            const uint32_t lp = before + incl - ll;
            if (lp + ll > regen || off == 0x1FFFFFFFu) bad = true; /* more literals than the block has; an offset beyond 29 bits */
            A.ll[q0 + i] = ll; A.ml[q0 + i] = ml; A.off[q0 + i] = off; A.lpos[q0 + i] = lbase + lp;
        }
        carry += s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    }
    if (carry > regen) bad = true;
    if (threadIdx.x == 0u && carry <= regen) { /* the block's last literals */
        A.ll[q0 + nseq] = regen - carry; A.ml[q0 + nseq] = 0u; A.off[q0 + nseq] = 0u; A.lpos[q0 + nseq] = lbase + carry;
    }
    if (bad) A.ok[f] = 0u;
}

/* bytes of every 256 flat sequences (k_lat_scan turns them into the workgroups' first output positions) */
__global__ void __launch_bounds__(256) k_zlat_sum(LatArgs A)
{
    __shared__ uint32_t s_sum[4];
    const uint32_t f = blockIdx.y, n = A.nseq[f];
    if (A.ok[f] == 0u || blockIdx.x * 256u >= n) return;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint64_t q = (uint64_t)f * A.nmax + i;
    const uint32_t len = i < n ? A.ll[q] + A.ml[q] : 0u;
    const uint32_t incl = scan64_incl(len);
    if ((threadIdx.x & 63u) == 63u) s_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0u) A.wgsum[f * (A.nmax / 256u) + blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

/* a thread per sequence: output position; a match starts inside the output so far */
__global__ void __launch_bounds__(256) k_zlat_place(LatArgs A)
{
    __shared__ uint32_t s_sum[4];
    const uint32_t f = blockIdx.y, n = A.nseq[f];
    if (A.ok[f] == 0u || blockIdx.x * 256u >= n) return;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool on = i < n;
    const uint64_t q = (uint64_t)f * A.nmax + i;
    uint32_t ll = 0, ml = 0;
    if (on) { ll = A.ll[q]; ml = A.ml[q]; }
    const uint32_t len = ll + ml, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t incl = scan64_incl(len);
    if (lane == 63u) s_sum[wv] = incl;
    __syncthreads();
    uint32_t before = A.wgsum[f * (A.nmax / 256u) + blockIdx.x];
    for (uint32_t w = 0; w < wv; w++) before += s_sum[w];
    const uint32_t op = before + incl - len;
    if (on) {
        A.opos[q] = op;
        const uint32_t off = A.off[q];
        if (ml != 0u && (off == 0u || off > op + ll)) A.ok[f] = 0u; /* k_zexec's rule: the offset may not reach before the output */
    }
}

} // namespace

/* ------------------------------------------------------------------------------------------- host */
namespace {

struct Layout {
    uint32_t F, nbmax, litcap, seqcap;
    size_t htmp_stride;
    size_t o_frames, o_blks, o_huf, o_seqt, o_predef, o_lits, o_seqs, o_chain, o_cnt, o_hitems, o_hitems2, o_htmp, o_irreg, o_sitems, o_hsegs, o_mitems, o_fused, total;
    /* few frames per call (k_zlat_*): sequence slots per frame, padded block size, jump rounds; 0 = the call is not one of those */
    uint32_t lat_nmax, lat_bpad, lat_rounds;
    size_t o_lat_nseq, o_lat_ok, o_lat_done, o_lat_zstart, o_lat_opos, o_lat_ll, o_lat_lpos, o_lat_ml, o_lat_off, o_lat_wgsum, o_lat_src, o_lat_changed;
};

/* calls the byte-parallel execution takes: what lz4_lat.hip takes (up to 64 blocks and 64 MiB per call) */
inline bool zstd_few_frames(uint64_t n_blocks, uint32_t B)
{
    return n_blocks >= 1u && n_blocks <= 64u && B >= (32u << 10) && B <= (2u << 20) && n_blocks * (uint64_t)B <= (64ull << 20);
}

constexpr uint32_t kFusedGridForIrregular = 256;
constexpr uint64_t kForkMaxZBlocks = 8192; /* calls of at most this many zstd blocks: one tile, its two entropy stages on two streams */

inline size_t al256(size_t v) { return (v + 255u) & ~(size_t)255u; }

Layout make_layout(uint64_t n_blocks, uint32_t B, size_t limit = ~(size_t)0)
{
    Layout y0;
    y0.nbmax = B / kZBlockMax + 2u;
    if (y0.nbmax > 254u) y0.nbmax = 254u;
    y0.litcap = ((B + 15u) & ~15u) + 32u * y0.nbmax;
    y0.htmp_stride = al256(2u * (size_t)y0.litcap + (size_t)y0.nbmax * kHwBlockSlack);
    const size_t per_frame = y0.htmp_stride + sizeof(ZFrame) + (size_t)y0.nbmax * (sizeof(ZBlk) + kHufTblWords * 2u + kSeqTblWords * 4u + 8u) +
                             y0.litcap + (size_t)(B / 6u) * (sizeof(uint2) + sizeof(uint2)) /* sequence pool share */ + 4u +
                             (size_t)y0.nbmax * (64u * sizeof(uint4) + 4u) /* k_zmove's segment lists */;
    /* Tile size.  K2 and K3 are bound by LDS capacity (two workgroups per CU, 512 per chip): 14848 frames
     * = 512 x 29 fill exactly one round of K3 (and 928 waves of 16 = two rounds of K2, the second 81 % full); the
     * workspace budget may force less. */
    static const size_t budget_env = cryo_tuning_env("CRYO_ZSTD_WS_MB") ? (size_t)atoll(cryo_tuning_env("CRYO_ZSTD_WS_MB")) << 20 : 0; /* tuning aid */
    const size_t budget = budget_env ? budget_env : (size_t)18 << 30; /* per tile in flight */
    uint64_t F = budget / per_frame;
    /* one full round of k_zchain: 512 waves x 29 zstd blocks; a frame of B bytes is ceil(B / 128 KiB) of them (round 3:
     * 1 MiB frames in tiles of 2320 made 1.25 rounds, the second three quarters empty) */
    /* (k_zchain4: 768 waves x 16 blocks, three waves per CU) */
    const uint64_t zbpf = (B + kZBlockMax - 1u) / kZBlockMax ? (B + kZBlockMax - 1u) / kZBlockMax : 1u;
    const uint64_t kTile = (768u * kCqW) / zbpf;
#ifndef CRYO_ZSTD_EQUAL_TILES
#define CRYO_ZSTD_EQUAL_TILES 1
#endif
    if (CRYO_ZSTD_EQUAL_TILES && n_blocks * zbpf > kForkMaxZBlocks) {
        /* Round 5: a call of more than one tile is cut into tiles of EQUAL size, four (the streams they run on) per round.  With
         * tiles of 12 288 zstd blocks 65 536 frames were 5.33 tiles: four in flight, then 1.33 with the chip half idle.  One
         * round of four larger tiles (up to 16 896 zstd blocks each, 17.6 GiB of workspace) instead: 65 536 x 128 KiB 330 ->
         * 335-350 GB/s, `narrow` 895 -> 950-965, 8 192 x 1 MiB 278 -> 313, 16 384 x 128 KiB 281 -> 292, level 5 134 -> 147
         * (profiles/r05_zstd_tiles*.txt; tiles of 928 ... 1 392 frames of 1 MiB lose, two tiles of 4 176 gain less).  From 8 193
         * zstd blocks on, i.e. where a call no longer runs its two entropy stages side by side (kForkMaxZBlocks; 12 288 frames as
         * four tiles 283 -> 302 GB/s; below that one tile with the fork is best: 8 192 frames 251, as four tiles 192) */
        const uint64_t kBig = (16384u + 512u) / zbpf ? (16384u + 512u) / zbpf : 1u;
        const uint64_t rounds = (n_blocks + 4u * kBig - 1u) / (4u * kBig);
        const uint64_t ntiles = 4u * rounds;
        uint64_t Fe = (n_blocks + ntiles - 1u) / ntiles;
        Fe = (Fe + 15u) & ~(uint64_t)15u;
        /* ... unless the tiles would be short of frames: k_zplan, k_zmat and k_zexec are a wave per FRAME (2 048 x 1 MiB as four
         * tiles of 512: 204 -> 190 GB/s; 4 096 x 1 MiB as four of 1 024: 261 -> 269) */
        if (Fe < 1024u) Fe = kTile;
        if (F > Fe) F = Fe;
    } else if (F > kTile) F = kTile;
    static const uint64_t tile_env = cryo_tuning_env("CRYO_ZSTD_TILE") ? (uint64_t)atoll(cryo_tuning_env("CRYO_ZSTD_TILE")) : 0; /* tuning aid (debug builds) */
    if (tile_env) F = tile_env;
    if (F < 16u) F = 16u;
    if (F > n_blocks) F = n_blocks;
    auto build = [&](const uint64_t Fq) -> Layout {
    Layout y = y0;
    y.F = (uint32_t)Fq;
    /* sequence records: a share of B/6 per frame (levels 4..9 reach B/8 on text-like rows; the format allows B/3:
     * frames that do not fit the pool go to the irregular list) */
    uint64_t seqcap = (uint64_t)y.F * (B / 6u) + 4096u;
    if (seqcap > 0xFFFF0000ull) seqcap = 0xFFFF0000ull;
    y.seqcap = (uint32_t)seqcap;
    size_t o = 0;
    y.o_frames = o; o = al256(o + (size_t)y.F * sizeof(ZFrame));
    y.o_blks = o; o = al256(o + (size_t)y.F * y.nbmax * sizeof(ZBlk));
    y.o_huf = o; o = al256(o + (size_t)y.F * y.nbmax * kHufTblWords * 2u);
    y.o_seqt = o; o = al256(o + (size_t)y.F * y.nbmax * kSeqTblWords * 4u);
    y.o_predef = o; o = al256(o + kSeqTblWords * 4u);
    y.o_lits = o; o = al256(o + (size_t)y.F * y.litcap + 64u);
    y.o_seqs = o; o = al256(o + (size_t)y.seqcap * sizeof(uint2));
    y.o_chain = o; o = al256(o + ((size_t)y.seqcap + 64u) * sizeof(uint2));
    y.o_cnt = o; o = al256(o + 256u);
    y.o_hitems = o; o = al256(o + (size_t)y.F * y.nbmax * 4u);
    y.o_hitems2 = o; o = al256(o + (size_t)y.F * y.nbmax * 4u);
    y.o_htmp = o; o = al256(o + (size_t)y.F * y.htmp_stride + 64u);
    y.o_irreg = o; o = al256(o + (size_t)y.F * 4u);
    y.o_sitems = o; o = al256(o + (size_t)y.F * y.nbmax * 4u);
    y.o_hsegs = o; o = al256(o + (size_t)y.F * y.nbmax * 64u * sizeof(uint4));
    y.o_mitems = o; o = al256(o + (size_t)y.F * y.nbmax * 4u);
    y.o_fused = o; o = al256(o + zstd_fused_workspace(kFusedGridForIrregular));
    y.lat_nmax = 0; y.lat_bpad = 0; y.lat_rounds = 0;
    if (zstd_few_frames(n_blocks, B) && y.F >= n_blocks) {
        const size_t n = (size_t)n_blocks;
        y.lat_nmax = (B / 4u + 2u * y.nbmax + 255u) & ~255u; /* the format allows B / 3 sequences: a frame beyond B / 4 stays with k_zexec */
        y.lat_bpad = (B + 4095u) & ~4095u;
        y.lat_rounds = 2;
        while ((1ull << (2u * (y.lat_rounds - 1u))) < B) y.lat_rounds++;
        const size_t per = n * y.lat_nmax * 4u;
        y.o_lat_nseq = o; o = al256(o + n * 4u);
        y.o_lat_ok = o; o = al256(o + n * 4u);
        y.o_lat_done = o; o = al256(o + n * 4u);
        y.o_lat_zstart = o; o = al256(o + n * y.nbmax * 4u);
        y.o_lat_opos = o; o = al256(o + per);
        y.o_lat_ll = o; o = al256(o + per);
        y.o_lat_lpos = o; o = al256(o + per);
        y.o_lat_ml = o; o = al256(o + per);
        y.o_lat_off = o; o = al256(o + per);
        y.o_lat_wgsum = o; o = al256(o + n * (y.lat_nmax / 256u) * 4u);
        y.o_lat_src = o; o = al256(o + n * (size_t)y.lat_bpad * 4u);
        y.o_lat_changed = o; o = al256(o + (y.lat_rounds + 1u) * 4u);
    }
    y.total = o;
    return y;
    };
    Layout y = build(F);
    /* a cap on the workspace (CRYO_OPT_WORKSPACE_MAX_BYTES) below one tile of that size: the largest tile that fits (at least
     * 16 frames, whatever the cap).  Largest, so that the launcher -- which is handed the workspace this was sized for, or a
     * larger one -- arrives at the same tile or a larger one, never at a remainder tile */
    if (limit != ~(size_t)0 && y.total > limit && F > 16u) {
        uint64_t lo = 16u, hi = F;
        while (lo < hi) {
            const uint64_t mid = (lo + hi + 1u) / 2u;
            if (build(mid).total <= limit) lo = mid; else hi = mid - 1u;
        }
        y = build(lo);
    }
    return y;
}

/* two tiles in flight on two side streams: the wave-per-frame kernels of one tile beside the entropy kernels of the
 * other (measured: 1 MiB blocks 95 -> 107 GB/s, where a tile has few frames; 128 KiB blocks 167 -> 173 GB/s,
 * profiles/r03_variants_ab.txt) */
int tile_lanes(uint32_t block_size)
{
    static const char *e = cryo_tuning_env("CRYO_ZSTD_LANES"); /* tuning aid: 1 .. kZstdLanes */
    if (e && e[0] >= '1' && e[0] <= '0' + kZstdLanes) return e[0] - '0';
    (void)block_size;
    return 4; /* measured at 128 KiB ... 1 MiB blocks, profiles/r03_zstd_tiles_in_flight.txt */
}

/* path: 0 automatic, 1 the fused one-wave-per-frame kernel, 2 the pipeline (CRYO_OPT_ZSTD_DECODE_PATH).  Automatic =
 * the pipeline for every batch: measured on 1 .. 64 blocks (profiles/r03_zstd_small_batches.txt) its lane-per-stream
 * entropy stages beat the fused kernel's wave-serial loops even for ONE frame (128 KiB: 5.5 against 8.0 ms, 1 MiB: 12.1
 * against 64.7 ms); the fused kernel decodes what the planner calls irregular. */
bool use_pipeline(int path) { return path != 1; }

} // namespace

size_t zstd_decompress_workspace(uint64_t n_blocks, uint32_t block_size, int path, size_t max_bytes)
{
    if (!use_pipeline(path)) return zstd_fused_workspace(n_blocks);
    const Layout y = make_layout(n_blocks, block_size, max_bytes == ~(size_t)0 ? max_bytes : (max_bytes > 256u ? max_bytes - 256u : 0u));
    const uint64_t nt = (n_blocks + y.F - 1u) / y.F;
    uint64_t nl = nt < (uint64_t)tile_lanes(block_size) ? nt : (uint64_t)tile_lanes(block_size);
    while (nl > 1u && (size_t)nl * y.total + 256u > max_bytes) nl--; /* fewer tiles in flight; the launcher takes what it is given */
    return (size_t)nl * y.total + 256;
}

hipError_t launch_zstd_decompress(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                                  const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                                  uint32_t block_size, uint64_t n_blocks, int32_t *d_status,
                                  void *d_workspace, size_t workspace_bytes, const ZstdAux *aux, int path)
{
    if (n_blocks == 0) return hipSuccess;
    if (!use_pipeline(path))
        return launch_zstd_fused(s, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size, n_blocks, d_status,
                                 d_workspace, workspace_bytes, nullptr, nullptr, 0);
    uint8_t *ws0 = (uint8_t *)(((uintptr_t)d_workspace + 255u) & ~(uintptr_t)255u);
    const Layout y = make_layout(n_blocks, block_size, workspace_bytes > (size_t)(ws0 - (uint8_t *)d_workspace) ? workspace_bytes - (size_t)(ws0 - (uint8_t *)d_workspace) : 0u);

    /* Tiles alternate between side streams with a workspace each (tile_lanes()): the kernels of one tile fill what
     * another tile's leave idle (k_zchain holds the whole LDS with two waves per CU, the tail of every kernel leaves CUs
     * empty).  65 536 x 128 KiB: 259 / 274 / 292 / 302 GB/s with 1 / 2 / 3 / 4 tiles in flight, no more beyond;
     * 8 192 x 1 MiB: 211 / 239 / 240 with 2 / 3 / 4. */
    const uint64_t ntiles = (n_blocks + y.F - 1u) / y.F;
    int nl = aux ? tile_lanes(block_size) : 1;
    if ((uint64_t)nl > ntiles) nl = (int)ntiles;
    {   /* as many tiles in flight as the workspace holds (zstd_decompress_workspace planned it under the caller's cap) */
        const size_t room = workspace_bytes - (size_t)(ws0 - (uint8_t *)d_workspace);
        if (workspace_bytes < (size_t)(ws0 - (uint8_t *)d_workspace) || room < y.total) return hipErrorInvalidValue;
        if ((size_t)nl * y.total > room) nl = (int)(room / y.total);
    }
    hipError_t e;
    if (nl > 1) {
        if ((e = hipEventRecord(aux->fork, s)) != hipSuccess) return e;
        for (int l = 0; l < nl; l++)
            if ((e = hipStreamWaitEvent(aux->lane[l], aux->fork, 0)) != hipSuccess) return e;
    }
    static const uint32_t huf_pad = cryo_tuning_env("CRYO_ZHUF_PAD") ? (uint32_t)atoi(cryo_tuning_env("CRYO_ZHUF_PAD")) : 0u; /* tuning aid: extra LDS to cap occupancy */
    static const uint32_t seq_pad = cryo_tuning_env("CRYO_ZSEQ_PAD") ? (uint32_t)atoi(cryo_tuning_env("CRYO_ZSEQ_PAD")) : 0u;
    static const bool want_stats = cryo_tuning_env("CRYO_ZSTD_STATS") != nullptr; /* debugging aid */
    static const bool old_huf = cryo_tuning_env("CRYO_ZHUF_OLD") != nullptr;
    /* Huffman blocks of fewer than 4 KiB of literals (run-dominated frames: `int4` has ~1 KiB per block) are not worth the
     * walkers' set-up (two-symbol table, marks, four phases): lane per stream decodes them, +8-12 % on `int4`, nothing lost
     * elsewhere; from 12 KiB on the walkers win (`narrow`, 10 KiB: -14 % with everything on k_zhuf; profiles/r05_hufw_min.txt) */
    static const uint32_t hufw_min = cryo_tuning_env("CRYO_ZHUFW_MIN") ? (uint32_t)atoi(cryo_tuning_env("CRYO_ZHUFW_MIN")) : 4096u;
    /* walkers per stream: one per 2 048 bits; 1 024 / 512 / 256 measured the same within the noise on `narrow`, `int4` and
     * level -5 streams, with the first blocks handed back for want of a meeting point (profiles/r05_hufw_seglog.txt) */
    static const uint32_t hufw_seglog = cryo_tuning_env("CRYO_ZHUFW_SEGLOG") ? (uint32_t)atoi(cryo_tuning_env("CRYO_ZHUFW_SEGLOG")) : 11u;
    static const uint32_t huf2_grid = cryo_tuning_env("CRYO_ZHUF2_GRID") ? (uint32_t)atoi(cryo_tuning_env("CRYO_ZHUF2_GRID")) : 512u;
    /* path 3 (CRYO_OPT_ZSTD_DECODE_PATH): the pipeline with k_zexec for every frame, i.e. without the few-frames execution
     * below -- how the tests run one call shape through both */
    const bool skip_lat = path == 3 || cryo_tuning_env("CRYO_ZSTD_NO_FEW") != nullptr;
    /* calls of at most this many zstd blocks (frames x blocks per frame) run a tile's two entropy stages side by side: 3-6 %
     * less time from 1 to 4 096 frames, 5 % MORE at a full tile of 12 288 (profiles/r05_zstd_fork.txt; tuning aid:
     * CRYO_ZSTD_FORK_ZBLOCKS) */
    static const uint64_t fork_max_zblocks = cryo_tuning_env("CRYO_ZSTD_FORK_ZBLOCKS") ? (uint64_t)atoll(cryo_tuning_env("CRYO_ZSTD_FORK_ZBLOCKS")) : kForkMaxZBlocks;
    uint64_t t = 0;
    for (uint64_t first = 0; first < n_blocks; first += y.F, t++) {
        const int l = (int)(t % (uint64_t)nl);
        hipStream_t st = nl > 1 ? aux->lane[l] : s;
        uint8_t *ws = ws0 + (size_t)l * y.total;
        ZPipe P;
        P.src_base = d_src; P.src_off = d_src_off; P.src_size = d_src_size;
        P.dst_base = d_dst; P.dst_stride = dst_stride; P.B = block_size;
        P.nbmax = y.nbmax; P.litcap = y.litcap; P.seqcap = y.seqcap;
        P.status = d_status;
        P.frames = (ZFrame *)(ws + y.o_frames);
        P.blks = (ZBlk *)(ws + y.o_blks);
        P.huf = (uint16_t *)(ws + y.o_huf);
        P.seqt = (uint32_t *)(ws + y.o_seqt);
        P.predef = (uint32_t *)(ws + y.o_predef);
        P.lits = ws + y.o_lits;
        P.seqs = (uint2 *)(ws + y.o_seqs);
        P.chain = (uint2 *)(ws + y.o_chain);
        P.counters = (uint32_t *)(ws + y.o_cnt);
        P.hitems = (uint32_t *)(ws + y.o_hitems);
        P.hitems2 = (uint32_t *)(ws + y.o_hitems2);
        P.htmp = ws + y.o_htmp;
        P.htmp_stride = y.htmp_stride;
        P.irregular = (uint32_t *)(ws + y.o_irreg);
        P.sitems = (uint32_t *)(ws + y.o_sitems);
        P.hsegs = (uint4 *)(ws + y.o_hsegs);
        {
            /* the few-frames execution below reads Huffman literals from the pool: the move stays for such calls */
            static const bool move_env = cryo_tuning_env("CRYO_ZSTD_MOVE") != nullptr; /* A/B aid: k_zmove for every block */
            P.in_place = (move_env || (y.lat_nmax != 0u && !skip_lat)) ? 0u : 1u;
        }
        P.mitems = (uint32_t *)(ws + y.o_mitems);
        P.done = nullptr;
        P.hufw_min = hufw_min;
        P.hufw_seglog = hufw_seglog;
        const uint64_t left = n_blocks - first;
        P.first = first;
        P.F = (uint32_t)(left < y.F ? left : y.F);
        if ((e = hipMemsetAsync(P.counters, 0, 256, st)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_zplan, dim3(P.F), dim3(64), 0, st, P);
        /* Inside a tile the Huffman stage and the sequence stage depend on k_zplan only and meet in k_zexec.  With many tiles
         * in flight other tiles' kernels fill what one stage leaves idle; a call that leaves the chip mostly empty (a read-ahead of a
         * few hundred blocks, ONE frame: the reference's own call shape, cache.c:178) ran them one after the other.  There
         * the sequence stage goes to a side stream (round 5, profiles/r05_zstd_fork.txt). */
        hipStream_t sq = st;
        const bool fork_stages = aux != nullptr && n_blocks * ((block_size + kZBlockMax - 1u) / kZBlockMax) <= (uint64_t)fork_max_zblocks;
        if (fork_stages) {
            sq = aux->side[l];
            if ((e = hipEventRecord(aux->planned[l], st)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(sq, aux->planned[l], 0)) != hipSuccess) return e;
        }
        const uint32_t zhuf_all = (P.F * P.nbmax + kHufPerWave - 1u) / kHufPerWave;
#ifndef CRYO_GS
#define CRYO_GS 7 /* grids sized for the chip: 1 k_zhuf, 2 k_zhufw, 4 k_zchain4 (0: one workgroup per descriptor slot, as in round 3) */
#endif
        /* ---- sequence stage (its own stream when forked; issued first so that it is not queued behind the Huffman kernels) ---- */
        auto seq_stage = [&](hipStream_t q) {
            const uint32_t zc_all = (P.F * P.nbmax + kCqW - 1u) / kCqW;
            hipLaunchKernelGGL(k_zchain4, dim3((CRYO_GS & 4) && zc_all > 768u ? 768u : zc_all), dim3(64), seq_pad, q, P);
            hipLaunchKernelGGL(k_zmat, dim3(P.F), dim3(64), 0, q, P);
        };
        if (fork_stages) {
            seq_stage(sq);
            if ((e = hipEventRecord(aux->seqs_done[l], sq)) != hipSuccess) return e;
        }
        /* ---- Huffman stage ---- */
        if (old_huf) hipLaunchKernelGGL(k_zhuf, dim3((CRYO_GS & 1) && zhuf_all > 1024u ? 1024u : zhuf_all), dim3(64), huf_pad, st, P, P.hitems, 1u);
        else {
            /* 1 792 of these are resident (seven per CU); a frame has one Huffman block per 128 KiB */
            const uint32_t zhufw_all = P.F * P.nbmax, zhufw_want = P.F * ((P.B + kZBlockMax - 1u) / kZBlockMax) < 3584u ? 3584u : P.F * ((P.B + kZBlockMax - 1u) / kZBlockMax);
            hipLaunchKernelGGL(k_zhufw, dim3((CRYO_GS & 2) && zhufw_all > zhufw_want ? zhufw_want : zhufw_all), dim3(64), huf_pad, st, P);
            hipLaunchKernelGGL(k_zmove, dim3(P.F * P.nbmax), dim3(64), 0, st, P);
            static const bool skip_fallbacks = cryo_tuning_env("CRYO_ZSTD_SKIP_FALLBACKS") != nullptr; /* timing experiment (debug builds): wrong if anything was handed back */
            if (!skip_fallbacks)
            hipLaunchKernelGGL(k_zhuf, dim3((CRYO_GS & 1) && zhuf_all > huf2_grid ? huf2_grid : zhuf_all), dim3(64), 0, st, P, P.hitems2, 61u); /* the walkers' hand-backs: rarely any */
        }
        if (fork_stages) {
            if ((e = hipStreamWaitEvent(st, aux->seqs_done[l], 0)) != hipSuccess) return e;
        } else {
            seq_stage(st);
        }
        if (y.lat_nmax != 0u && !skip_lat) { /* few frames: every output byte in parallel; k_zexec takes what this leaves */
            LatArgs A = {};
            A.src_base = d_src; A.src_off = d_src_off + first; A.src_size = d_src_size + first;
            A.dst_base = d_dst + first * dst_stride; A.dst_stride = dst_stride; A.B = block_size; A.n_blocks = P.F; A.status = d_status + first;
            A.nmax = y.lat_nmax; A.bpad = y.lat_bpad;
            A.nseq = (uint32_t *)(ws + y.o_lat_nseq); A.ok = (uint32_t *)(ws + y.o_lat_ok); A.done = (uint32_t *)(ws + y.o_lat_done);
            A.opos = (uint32_t *)(ws + y.o_lat_opos); A.ll = (uint32_t *)(ws + y.o_lat_ll); A.lpos = (uint32_t *)(ws + y.o_lat_lpos);
            A.ml = (uint32_t *)(ws + y.o_lat_ml); A.off = (uint32_t *)(ws + y.o_lat_off);
            A.wgsum = (uint32_t *)(ws + y.o_lat_wgsum); A.src = (uint32_t *)(ws + y.o_lat_src); A.changed = (uint32_t *)(ws + y.o_lat_changed);
            A.pool_base = P.lits; A.pool_stride = P.litcap;
            uint32_t *zstart = (uint32_t *)(ws + y.o_lat_zstart);
            if ((e = hipMemsetAsync(A.changed, 0, (y.lat_rounds + 1u) * 4u, st)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_zlat_count, dim3(P.F), dim3(64), 0, st, P, A, zstart);
            hipLaunchKernelGGL(k_zlat_build, dim3(P.nbmax, P.F), dim3(256), 0, st, P, A, zstart);
            hipLaunchKernelGGL(k_zlat_sum, dim3(y.lat_nmax / 256u, P.F), dim3(256), 0, st, A);
            hipLaunchKernelGGL(k_lat_scan, dim3(P.F), dim3(64), 0, st, A);
            hipLaunchKernelGGL(k_zlat_place, dim3(y.lat_nmax / 256u, P.F), dim3(256), 0, st, A);
            hipLaunchKernelGGL(k_lat_fill, dim3((block_size + 4095u) / 4096u, P.F), dim3(256), 0, st, A);
            for (uint32_t r = 0; r < y.lat_rounds; r++)
                hipLaunchKernelGGL(k_lat_jump, dim3((block_size + 1023u) / 1024u, P.F), dim3(256), 0, st, A, r);
            hipLaunchKernelGGL(k_lat_gather, dim3((block_size + 4095u) / 4096u, P.F), dim3(256), 0, st, A);
            P.done = A.done;
        }
        hipLaunchKernelGGL(k_zexec, dim3(P.F), dim3(64), 0, st, P);
        const uint64_t fg = P.F < kFusedGridForIrregular ? P.F : kFusedGridForIrregular;
        static const bool skip_fused = cryo_tuning_env("CRYO_ZSTD_SKIP_FALLBACKS") != nullptr;
        if (!skip_fused)
        e = launch_zstd_fused(st, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size, fg, d_status,
                              ws + y.o_fused, zstd_fused_workspace(kFusedGridForIrregular), P.irregular,
                              P.counters + 2, first);
        if (e != hipSuccess) return e;
        if (want_stats) {
            uint32_t h[64];
            (void)hipMemcpyAsync(h, P.counters, sizeof h, hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            fprintf(stderr, "[zstd pipe] tile %llu: frames %u seqs %u huf items %u (handed back by the walkers: %u) irregular %u | huf log histogram:",
                    (unsigned long long)first, P.F, h[0], h[1], h[61], h[2]);
            for (int k = 1; k <= 12; k++) fprintf(stderr, " %d:%u", k, h[4 + k]);
            fprintf(stderr, " | LL log:");
            for (int k = 0; k <= 9; k++) fprintf(stderr, " %d:%u", k, h[20 + k]);
            fprintf(stderr, " | OF log:");
            for (int k = 0; k <= 9; k++) fprintf(stderr, " %d:%u", k, h[32 + k]);
            fprintf(stderr, " | ML log:");
            for (int k = 0; k <= 9; k++) fprintf(stderr, " %d:%u", k, h[44 + k]);
            fprintf(stderr, "\n[zstd pipe] k_zhufw wave time, units of 64 memtime ticks summed over waves: setup %u head %u body %u tail %u resolve+copy %u\n[zstd pipe] k_zplan: literals section (Huffman table) %u, sequence headers %u, FSE tables %u, table copies %u\n[zstd pipe] handed back because: table log %u, stream without end mark %u, no meeting point / overrun %u, counts do not add up %u, scratch full %u, skip > n %u\n", h[56], h[57], h[58], h[59], h[62], h[17], h[18], h[19], h[63], h[30], h[31], h[42], h[43], h[54], h[55]);
        }
    }
    if (nl > 1) {
        for (int l = 0; l < nl; l++) {
            if ((e = hipEventRecord(aux->join[l], aux->lane[l])) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(s, aux->join[l], 0)) != hipSuccess) return e;
        }
    }
    return hipGetLastError();
}

} // namespace cryo
