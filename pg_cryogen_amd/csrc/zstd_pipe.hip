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
 *   K1 k_zplan  wave per frame   frame/block/section headers -> descriptors; Huffman and FSE decoding
 *                                tables built in LDS, stored to the workspace
 *   K2 k_zhuf   lane per stream  64 Huffman streams per wave (16 blocks x 4 streams), the 16 tables in
 *                                LDS; symbols go to the frame's literal pool
 *   K3 k_zseq   lane per frame   FSE sequence decode incl. repeat offsets -> (ll, ml, offset) records
 *   K4 k_zexec  wave per frame   sequence execution with the shared LZ copy engine (lz_common.h):
 *                                records are loaded 64 at a time, literals stream through the LDS
 *                                input ring, output ring in LDS, 1 KiB coalesced flushes
 *
 * Anything K1 does not recognise as "one well-formed frame of at most nbmax blocks" (concatenated or
 * skippable frames, malformed headers, pool exhaustion) is put on an irregular list and decoded by the
 * fused kernel afterwards, so coverage and error behaviour are exactly the fused decoder's.
 *
 * The batch is processed in tiles of F frames so the workspace stays bounded (~4 GiB).
 */
#include "zstd_common.h"
#include "kernels.h"
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
    uint32_t pad[9];
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
    uint4 *seqs;
    uint32_t *counters; /* [0] sequence pool cursor, [1] Huffman items, [2] irregular frames */
    uint32_t *hitems;
    uint32_t *irregular;
};

struct PlanLds {
    uint16_t huf[1 << kHufLogMax];
    uint32_t ll[512], ml[512], of[256];
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

/* K1: one compressed block's section headers -> descriptor + tables.  false = not plannable. */
__device__ bool plan_block(PlanLds &L, const ZPipe &P, PlanState &ps, const uint8_t *src, uint32_t n, uint32_t boff,
                           uint32_t f, uint32_t k, uint32_t &lit_cursor, ZBlk &d, uint32_t lane)
{
    if (n < 3u) return false;
    const uint32_t b0 = uni(src[0]);
    const uint32_t type = b0 & 3u, fmt = (b0 >> 2) & 3u;
    uint32_t regen, used;
    d.nstreams = 0;
    d.huf_slot = 0; d.huf_log = 0;
    if (type < 2u) {
        uint32_t hdr;
        if (fmt == 1u) { hdr = 2; regen = (b0 >> 4) | (uni(src[1]) << 4); }
        else if (fmt == 3u) { hdr = 3; regen = (b0 >> 4) | (uni(src[1]) << 4) | (uni(src[2]) << 12); }
        else { hdr = 1; regen = b0 >> 3; }
        if (type == 0u) {
            if (hdr + regen > n || regen > kZBlockMax) return false;
            d.lit_mode = 0; d.lit_src = boff + hdr; used = hdr + regen;
        } else {
            if ((fmt == 3u && n < 4u) || regen > kZBlockMax || hdr + 1u > n) return false;
            d.lit_mode = 1; d.lit_src = uni(src[hdr]); used = hdr + 1u;
        }
    } else {
        if (n < 5u) return false;
        const uint32_t h = b0 | (uni(src[1]) << 8) | (uni(src[2]) << 16) | (uni(src[3]) << 24);
        uint32_t hdr, csize;
        bool single = false;
        if (fmt < 2u) { single = (fmt == 0u); hdr = 3; regen = (h >> 4) & 0x3FFu; csize = (h >> 14) & 0x3FFu; }
        else if (fmt == 2u) { hdr = 4; regen = (h >> 4) & 0x3FFFu; csize = h >> 18; }
        else { hdr = 5; regen = (h >> 4) & 0x3FFFFu; csize = (h >> 22) + (uni(src[4]) << 10); }
        if (regen > kZBlockMax || csize + hdr > n) return false;
        const uint8_t *p = src + hdr;
        uint32_t left = csize;
        if (type == 3u) { if (!ps.huf_valid) return false; }
        else {
            int hlog = 0;
            const int t = huf_read_table(L, p, left, &hlog, lane);
            if (t < 0) return false;
            ps.huf_valid = true;
            ps.huf_slot = k;
            ps.huf_log = (uint32_t)hlog;
            copy_words(reinterpret_cast<uint32_t *>(P.huf + ((uint64_t)f * P.nbmax + k) * kHufTblWords),
                       reinterpret_cast<const uint32_t *>(L.huf), (1u << hlog) >> 1, lane);
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
            const uint32_t l1 = uni((uint32_t)p[0] | ((uint32_t)p[1] << 8));
            const uint32_t l2 = uni((uint32_t)p[2] | ((uint32_t)p[3] << 8));
            const uint32_t l3 = uni((uint32_t)p[4] | ((uint32_t)p[5] << 8));
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
        lit_cursor += (regen + 15u) & ~15u;
        used = hdr + csize;
    }
    d.regen = regen;

    /* sequences section header */
    const uint8_t *ip = src + used;
    uint32_t left = n - used;
    if (left < 1u) return false;
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
#pragma unroll
        for (int kind = 0; kind < 3; kind++) { /* LL, OF, ML in stream order */
            const int mode = (int)((modes >> (6 - 2 * kind)) & 3u);
            uint32_t *lt = kind == 0 ? L.ll : (kind == 1 ? L.of : L.ml);
            const uint32_t goff = kind == 0 ? 0u : (kind == 1 ? 1024u : 512u);
            if (mode == 0) { /* predefined: shared table built once per tile */
                ps.slot[kind] = kPredefSlot;
                ps.log[kind] = kind == 1 ? 5u : 6u;
            } else {
                int lg = 0;
                const int u = read_seq_table(L, lt, &lg, kind, mode, ip, left, ps.fse_valid);
                if (u < 0) return false;
                ip += u; left -= (uint32_t)u;
                if (mode != 3) {
                    __builtin_amdgcn_wave_barrier();
                    copy_words(gt + goff, lt, 1u << lg, lane);
                    __builtin_amdgcn_wave_barrier();
                    ps.slot[kind] = k;
                    ps.log[kind] = (uint32_t)lg;
                }
            }
        }
        ps.fse_valid = true;
        d.sq_off = boff + (uint32_t)(ip - src);
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
            copy_words(P.predef + (kind == 0 ? 0u : (kind == 1 ? 1024u : 512u)), lt, 1u << lg, lane);
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
        const uint32_t fhd = uni(src[4]);
        const uint32_t single = (fhd >> 5) & 1u, did = fhd & 3u, fcs_flag = fhd >> 6, has_ck = (fhd >> 2) & 1u;
        const uint32_t did_sz = did == 3u ? 4u : did;
        const uint32_t fcs_sz = fcs_flag == 0u ? single : (1u << fcs_flag);
        const uint32_t hsz = 5u + (single ? 0u : 1u) + did_sz + fcs_sz;
        if ((fhd & 0x08u) || csize < hsz) break;
        uint32_t p = 5u;
        if (!single) { if ((uni(src[p]) >> 3) + 10u > 31u) break; p++; }
        if (did) {
            uint32_t id = 0;
            for (uint32_t k = 0; k < did_sz; k++) id |= uni(src[p + k]) << (8u * k);
            if (id != 0u) break;
            p += did_sz;
        }
        uint64_t fcs = ~0ull;
        if (fcs_flag == 0u) { if (single) fcs = uni(src[p]); }
        else {
            uint64_t v = 0;
            for (uint32_t k = 0; k < fcs_sz; k++) v |= (uint64_t)uni(src[p + k]) << (8u * k);
            fcs = fcs_flag == 1u ? v + 256u : v;
        }
        uint32_t ip = hsz;
        PlanState ps = {};
        uint32_t lit_cursor = 0;
        bool okf = true;
        for (;;) {
            if (nblk >= P.nbmax || csize - ip < 3u) { okf = false; break; }
            const uint32_t bh = uni((uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16));
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
                    if (!plan_block(L, P, ps, src + ip, bsize, ip, f, nblk, lit_cursor, d, lane)) { okf = false; break; }
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
                if (bd[k].type == 2u && bd[k].lit_mode == 2u) P.hitems[atomicAdd(&P.counters[1], 1u)] = f * P.nbmax + k;
        } else {
            fr.flags = F_IRREG;
            P.irregular[atomicAdd(&P.counters[2], 1u)] = f;
        }
        fr.nblk = nblk;
        P.frames[f] = fr;
    }
}

/* ------------------------------------------------------------------------------------------------ K2 */
constexpr uint32_t kHufPerWave = 16; /* blocks per wave: 16 x 4 streams = 64 lanes */
constexpr uint32_t kHufLdsLog = 11;  /* tables up to 2^11 entries sit in LDS; 2^12 ones are read from L2 */

__global__ void __launch_bounds__(64) k_zhuf(ZPipe P)
{
    __shared__ __attribute__((aligned(16))) uint16_t tbl[kHufPerWave << kHufLdsLog];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nitems = uni(P.counters[1]);
    const uint32_t base = blockIdx.x * kHufPerWave;
    if (base >= nitems) return;

    /* stage the 16 tables: 4 KiB each, 16 bytes per lane per step */
    for (uint32_t j = 0; j < kHufPerWave && base + j < nitems; j++) {
        const uint32_t it = uni(P.hitems[base + j]);
        const ZBlk *d = P.blks + it;
        const uint32_t f = it / P.nbmax;
        const uint32_t hlog = uni(d->huf_log);
        if (hlog > kHufLdsLog) continue;
        const uint4 *g = reinterpret_cast<const uint4 *>(P.huf + ((uint64_t)f * P.nbmax + uni(d->huf_slot)) * kHufTblWords);
        uint4 *l = reinterpret_cast<uint4 *>(tbl + (j << kHufLdsLog));
        const uint32_t n16 = ((2u << hlog) + 15u) >> 4;
        for (uint32_t i = lane; i < n16; i += 64u) l[i] = g[i];
    }
    __builtin_amdgcn_wave_barrier();

    const uint32_t j = lane >> 2, sid = lane & 3u;
    if (base + j >= nitems) return;
    const uint32_t it = P.hitems[base + j];
    const ZBlk *d = P.blks + it;
    const uint32_t f = it / P.nbmax;
    const uint32_t nstreams = d->nstreams;
    if (sid >= nstreams) return;
    const uint32_t regen = d->regen, hlog = d->huf_log;
    const uint32_t seg = (regen + 3u) / 4u;
    uint32_t cnt, oofs;
    if (nstreams == 1u) { cnt = regen; oofs = 0; }
    else { cnt = sid < 3u ? seg : regen - 3u * seg; oofs = sid * seg; }
    const uint64_t blk = P.first + f;
    const uint8_t *src = P.src_base + P.src_off[blk];
    uint8_t *o = P.lits + (uint64_t)f * P.litcap + d->lit_src + oofs;

    BitRd b;
    bool ok = b.init(src + d->hs_off[sid], d->hs_len[sid]);
    if (ok) {
        uint32_t i = 0;
        if (hlog <= kHufLdsLog) {
            const uint16_t *t = tbl + (j << kHufLdsLog);
            for (; i + 8u <= cnt; i += 8u) { /* 8 symbols -> one 8-byte store */
                uint64_t acc = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint32_t e = t[b.peek(hlog)];
                    acc |= (uint64_t)(e & 255u) << (8 * q);
                    b.skip(e >> 8);
                }
                __builtin_memcpy(o + i, &acc, 8);
            }
            for (; i < cnt; i++) {
                const uint32_t e = t[b.peek(hlog)];
                o[i] = (uint8_t)e;
                b.skip(e >> 8);
            }
        } else {
            const uint16_t *t = P.huf + ((uint64_t)f * P.nbmax + d->huf_slot) * kHufTblWords;
            for (; i < cnt; i++) {
                const uint32_t e = t[b.peek(hlog)];
                o[i] = (uint8_t)e;
                b.skip(e >> 8);
            }
        }
        ok = (b.pos == 0) && !b.over; /* must end exactly */
    }
    if (!ok) atomicOr(&P.frames[f].flags, F_BAD);
}

/* ------------------------------------------------------------------------------------------------ K3 */
__global__ void __launch_bounds__(64) k_zseq(ZPipe P)
{
    const uint32_t f = blockIdx.x * 64u + (threadIdx.x & 63u);
    if (f >= P.F) return;
    const uint32_t flags = P.frames[f].flags;
    if (flags & (F_BAD | F_IRREG)) return;
    const uint32_t nblk = P.frames[f].nblk;
    const uint64_t blk = P.first + f;
    const uint8_t *src = P.src_base + P.src_off[blk];
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8;
    bool bad = false;
    for (uint32_t k = 0; k < nblk && !bad; k++) {
        const ZBlk *d = P.blks + (uint64_t)f * P.nbmax + k;
        if (d->type != 2u) continue;
        const uint32_t nseq = d->nseq;
        if (nseq == 0u) continue;
        const uint32_t slots = d->slots, logs = d->logs;
        const uint32_t ll_log = logs & 255u, of_log = (logs >> 8) & 255u, ml_log = (logs >> 16) & 255u;
        auto tab = [&](uint32_t slot, uint32_t goff) -> const uint32_t * {
            return (slot == kPredefSlot ? P.predef : P.seqt + ((uint64_t)f * P.nbmax + slot) * kSeqTblWords) + goff;
        };
        const uint32_t *tl = tab(slots & 255u, 0u), *to = tab((slots >> 8) & 255u, 1024u), *tm = tab((slots >> 16) & 255u, 512u);
        uint4 *out = P.seqs + d->seq_base;
        BitRd b;
        if (!b.init(src + d->sq_off, d->sq_len)) { bad = true; break; }
        uint32_t sl = b.read(ll_log);
        uint32_t so = b.read(of_log);
        uint32_t sm = b.read(ml_log);
        for (uint32_t i = 0; i < nseq; i++) {
            const uint32_t el = tl[sl], eo = to[so], em = tm[sm];
            const uint32_t lsym = el >> 14, osym = eo >> 14, msym = em >> 14;
            const uint32_t llbase = kLLBase[lsym], llbits = kLLBits[lsym];
            const uint32_t mlbase = kMLBase[msym], mlbits = kMLBits[msym];
            const bool ll0 = (llbase == 0u);
            uint32_t offset;
            if (osym > 1u) {
                offset = ((1u << osym) - 3u) + b.read(osym);
                rep2 = rep1; rep1 = rep0; rep0 = offset;
            } else if (osym == 0u) {
                if (!ll0) offset = rep0;
                else { offset = rep1; rep1 = rep0; rep0 = offset; }
            } else {
                const uint32_t idx = 1u + (ll0 ? 1u : 0u) + b.read(1u);
                uint32_t tmp = (idx == 3u) ? rep0 - 1u : (idx == 1u ? rep1 : rep2);
                if (tmp == 0u) tmp = 1u;
                if (idx != 1u) rep2 = rep1;
                rep1 = rep0;
                rep0 = offset = tmp;
            }
            const uint32_t mlen = mlbase + (mlbits ? b.read(mlbits) : 0u);
            const uint32_t llen = llbase + (llbits ? b.read(llbits) : 0u);
            if (i + 1u < nseq) {
                sl = (el & 1023u) + b.read((el >> 10) & 15u);
                sm = (em & 1023u) + b.read((em >> 10) & 15u);
                so = (eo & 1023u) + b.read((eo >> 10) & 15u);
            }
            out[i] = make_uint4(llen, mlen, offset, 0u);
        }
        if (b.over || b.pos != 0) bad = true; /* the bitstream must be consumed exactly */
    }
    if (bad) atomicOr(&P.frames[f].flags, F_BAD);
}

/* ------------------------------------------------------------------------------------------------ K4 */
namespace {

struct ExecLds {
    uint8_t ring[ZR];
    uint8_t in[kInRing];
    unsigned long long meta[64];
    uint32_t bm[kTMax / 32];
};

/* execute one compressed block's sequences; false on malformed input */
__device__ bool exec_block(ExecLds &L, Wave<ZR> &w, const ZPipe &P, const ZBlk *d, const uint8_t *src, uint32_t f,
                           uint32_t cap, uint32_t lane, Stats &st)
{
    const uint32_t regen = uni(d->regen), nseq = uni(d->nseq);
    const int lit_mode = (int)uni(d->lit_mode);
    const uint32_t lit_src = uni(d->lit_src);
    const uint32_t rle_byte = lit_src;
    uint32_t lit_pos = 0, lvp = 0;
    if (lit_mode == 0) lvp = stream_open(w, src + lit_src, regen);
    else if (lit_mode == 2) lvp = stream_open(w, P.lits + (uint64_t)f * P.litcap + lit_src, regen);
    const bool streamed = lit_mode != 1;

    const uint4 *seqs = P.seqs + uni(d->seq_base);
    uint32_t q_ll = 0, q_ml = 0, q_off = 0;
    uint32_t qn = 0, loaded = 0;
    uint4 nx = make_uint4(0, 0, 0, 0); /* records loaded .. loaded+63, one per lane, in flight */
    if (lane < nseq) nx = seqs[lane];
    for (;;) {
        /* ---- top the queue up from the prefetched records ---- */
        if (qn < 64u && loaded < nseq) {
            uint32_t take = 64u - qn;
            if (take > nseq - loaded) take = nseq - loaded;
            const int from = (int)((lane - qn) & 63u);
            const uint32_t a = (uint32_t)__shfl((int)nx.x, from, 64), b2 = (uint32_t)__shfl((int)nx.y, from, 64),
                           c = (uint32_t)__shfl((int)nx.z, from, 64);
            if (lane >= qn && lane < qn + take) { q_ll = a; q_ml = b2; q_off = c; }
            qn += take;
            loaded += take;
            nx = make_uint4(0, 0, 0, 0);
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
        const bool ok = inq && streamed && q_ml <= q_off && q_off <= mabs && q_off < (1u << 21) &&
                        litend <= regen - lit_pos && oend <= kTMax && (uint64_t)w.op + oend <= cap;
        const unsigned long long badmask = __ballot(!ok);
        const uint32_t nb = badmask ? ctz64(badmask) : 64u;
        if (nb > 0u) {
            const uint32_t T = lane_get(oend, nb - 1u);
            const uint32_t lits = lane_get(litend, nb - 1u);
            while (w.in_hi < w.vend && w.in_hi < lvp + lits + 8u) w.refill();
            batch_copy<ZR>(w, L.in, L.meta, L.bm, nb, ostart, q_ll, q_off, (lvp + (litend - q_ll)) - ostart, T, st);
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

__global__ void __launch_bounds__(64) k_zexec(ZPipe P)
{
    __shared__ __attribute__((aligned(16))) ExecLds L;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t f = blockIdx.x;
    const uint32_t flags = uni(P.frames[f].flags);
    if (flags & F_IRREG) return; /* the fused decoder writes this block and its status */
    const uint64_t blk = P.first + f;
    if (flags & F_BAD) {
        if (lane == 0) P.status[blk] = CRYO_ST_CORRUPT;
        return;
    }
    const uint8_t *src = P.src_base + uni64(P.src_off[blk]);
    const uint32_t nblk = uni(P.frames[f].nblk);
    const uint32_t B = P.B;
    Stats st = {};
    Wave<ZR> w;
    w.ring = L.ring;
    w.in = L.in;
    w.lane = lane;
    w.dst = P.dst_base + uni64(blk * P.dst_stride);
    w.dst_aligned = (reinterpret_cast<uintptr_t>(w.dst) & 15u) == 0;
    w.op = 0;
    w.flushed = 0;
    w.delta = 0; w.abase = src; w.vend = 0; w.in_hi = 0; w.pre = make_uint2(0, 0);

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

/* ------------------------------------------------------------------------------------------- host */
namespace {

struct Layout {
    uint32_t F, nbmax, litcap, seqcap;
    size_t o_frames, o_blks, o_huf, o_seqt, o_predef, o_lits, o_seqs, o_cnt, o_hitems, o_irreg, o_fused, total;
};

constexpr uint32_t kFusedGridForIrregular = 256;

inline size_t al256(size_t v) { return (v + 255u) & ~(size_t)255u; }

Layout make_layout(uint64_t n_blocks, uint32_t B)
{
    Layout y;
    y.nbmax = B / kZBlockMax + 2u;
    if (y.nbmax > 254u) y.nbmax = 254u;
    y.litcap = ((B + 15u) & ~15u) + 16u * y.nbmax;
    const size_t per_frame = sizeof(ZFrame) + (size_t)y.nbmax * (sizeof(ZBlk) + kHufTblWords * 2u + kSeqTblWords * 4u + 4u) +
                             y.litcap + (size_t)B /* sequence pool share: B/16 records */ + 4u;
    const size_t budget = (size_t)4 << 30;
    uint64_t F = budget / per_frame;
    if (F < 64u) F = 64u;
    if (F > n_blocks) F = n_blocks;
    if (F > (1u << 20)) F = 1u << 20;
    y.F = (uint32_t)F;
    uint64_t seqcap = (uint64_t)y.F * (B / 16u) + 4096u;
    if (seqcap > 0xFFFF0000ull) seqcap = 0xFFFF0000ull;
    y.seqcap = (uint32_t)seqcap;
    size_t o = 0;
    y.o_frames = o; o = al256(o + (size_t)y.F * sizeof(ZFrame));
    y.o_blks = o; o = al256(o + (size_t)y.F * y.nbmax * sizeof(ZBlk));
    y.o_huf = o; o = al256(o + (size_t)y.F * y.nbmax * kHufTblWords * 2u);
    y.o_seqt = o; o = al256(o + (size_t)y.F * y.nbmax * kSeqTblWords * 4u);
    y.o_predef = o; o = al256(o + kSeqTblWords * 4u);
    y.o_lits = o; o = al256(o + (size_t)y.F * y.litcap + 64u);
    y.o_seqs = o; o = al256(o + (size_t)y.seqcap * sizeof(uint4));
    y.o_cnt = o; o = al256(o + 64u);
    y.o_hitems = o; o = al256(o + (size_t)y.F * y.nbmax * 4u);
    y.o_irreg = o; o = al256(o + (size_t)y.F * 4u);
    y.o_fused = o; o = al256(o + zstd_fused_workspace(kFusedGridForIrregular));
    y.total = o;
    return y;
}

bool use_pipeline(uint64_t n_blocks)
{
    static const char *e = getenv("CRYO_ZSTD_PIPE"); /* 0 = always fused, 1 = always pipeline (testing) */
    if (e && e[0] == '0') return false;
    if (e && e[0] == '1') return true;
    return n_blocks >= 16u;
}

} // namespace

size_t zstd_decompress_workspace(uint64_t n_blocks, uint32_t block_size)
{
    if (!use_pipeline(n_blocks)) return zstd_fused_workspace(n_blocks);
    return make_layout(n_blocks, block_size).total + 256;
}

hipError_t launch_zstd_decompress(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                                  const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                                  uint32_t block_size, uint64_t n_blocks, int32_t *d_status,
                                  void *d_workspace, size_t workspace_bytes)
{
    if (n_blocks == 0) return hipSuccess;
    if (!use_pipeline(n_blocks))
        return launch_zstd_fused(s, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size, n_blocks, d_status,
                                 d_workspace, workspace_bytes, nullptr, nullptr, 0);
    const Layout y = make_layout(n_blocks, block_size);
    if (workspace_bytes < y.total) return hipErrorInvalidValue;
    uint8_t *ws = (uint8_t *)(((uintptr_t)d_workspace + 255u) & ~(uintptr_t)255u);
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
    P.seqs = (uint4 *)(ws + y.o_seqs);
    P.counters = (uint32_t *)(ws + y.o_cnt);
    P.hitems = (uint32_t *)(ws + y.o_hitems);
    P.irregular = (uint32_t *)(ws + y.o_irreg);
    for (uint64_t first = 0; first < n_blocks; first += y.F) {
        const uint64_t left = n_blocks - first;
        P.first = first;
        P.F = (uint32_t)(left < y.F ? left : y.F);
        hipError_t e = hipMemsetAsync(P.counters, 0, 64, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_zplan, dim3(P.F), dim3(64), 0, s, P);
        hipLaunchKernelGGL(k_zhuf, dim3((P.F * P.nbmax + kHufPerWave - 1u) / kHufPerWave), dim3(64), 0, s, P);
        hipLaunchKernelGGL(k_zseq, dim3((P.F + 63u) / 64u), dim3(64), 0, s, P);
        hipLaunchKernelGGL(k_zexec, dim3(P.F), dim3(64), 0, s, P);
        const uint64_t fg = P.F < kFusedGridForIrregular ? P.F : kFusedGridForIrregular;
        e = launch_zstd_fused(s, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size, fg, d_status,
                              ws + y.o_fused, zstd_fused_workspace(kFusedGridForIrregular), P.irregular,
                              P.counters + 2, first);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

} // namespace cryo
