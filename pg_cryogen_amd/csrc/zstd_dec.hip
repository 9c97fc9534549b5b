/*
 * zstd_dec.hip -- Zstandard frame decode, one wavefront per cryo block.
 *
 * Replaces ZSTD_decompress(out, CRYO_BLCKSZ, compressed, compressed_size)
 * (reference compression.c:116) for a batch of independent blocks.  A 128 KiB cryo block is
 * one zstd block; the reference's 1 MiB block is a frame of 8 dependent blocks (repeat
 * offsets, repeat-mode entropy tables and the window carry over), decoded in order by the
 * same wave.
 *
 * Format coverage: concatenated and skippable frames; raw / RLE / compressed blocks; raw /
 * RLE / Huffman (1 or 4 streams) / treeless literals; predefined / RLE / FSE / repeat
 * sequence tables; repeat offsets; optional XXH64 content checksum.  Structural checks are
 * those of libzstd 1.4.8's one-shot decoder; every entropy bitstream must be consumed exactly
 * (RFC 8878), which is stricter than libzstd in two undefined-read corners.
 *
 * Work split inside the wave (64-thread workgroup, grid-stride over blocks):
 *   - table descriptions, FSE weight decode, FSE sequence decode: wave-uniform (serial by
 *     nature: one adaptive bitstream);
 *   - Huffman literals: the 4 streams are decoded by lanes 0..3 in parallel (table in LDS),
 *     into a per-workgroup literal buffer in HBM/L2 that the sequence stage streams back
 *     through the LDS input ring;
 *   - sequence execution: the 64 lanes co-operate on every literal run and match exactly as
 *     the LZ4 decoder does (lz_common.h): output ring in LDS, near matches LDS->LDS, far
 *     matches read back from the flushed output, 1 KiB coalesced flushes to HBM.
 */
#include "zstd_common.h"
#include "lz4_copy.h"
#include <cstdio>
#include <cstdlib>

namespace cryo {

namespace {

struct ZLds {
    uint8_t ring[ZR + 16];   /* + the copy engine's 16-byte tail (lz4_copy.h) */
    uint8_t in[kInRing + 16];
    uint16_t huf[1 << kHufLogMax]; /* symbol | nbits << 8 */
    uint32_t ll[512], ml[512], of[256]; /* next | nbits << 10 | symbol << 14 */
    int16_t norm[256];
    uint16_t nxt[256];
    uint32_t wdt[64]; /* FSE table of the Huffman weights */
    uint8_t wts[256];
    uint8_t cell[512];
    uint32_t meta[64];           /* copy engine: match meta */
    uint32_t bm[CopyLds<ZR, kTMax>::kWords]; /* copy engine: bitmap of match starts + per-chunk bases */
};

/* decode the 4 (or 1) Huffman streams with lanes 0..3; symbols go to the literal buffer */
__device__ bool huf_decode_streams(const ZLds &L, int hlog, uint8_t *lit, uint32_t regen, const uint8_t *p,
                                   uint32_t left, bool single, uint32_t lane)
{
    uint32_t sofs[4] = {0, 0, 0, 0}, slen[4] = {left, 0, 0, 0}, cnt[4] = {regen, 0, 0, 0}, oofs[4] = {0, 0, 0, 0};
    if (!single) {
        if (left < 10u) return false;
        const uint32_t l1 = uni((uint32_t)p[0] | ((uint32_t)p[1] << 8));
        const uint32_t l2 = uni((uint32_t)p[2] | ((uint32_t)p[3] << 8));
        const uint32_t l3 = uni((uint32_t)p[4] | ((uint32_t)p[5] << 8));
        if (6u + l1 + l2 + l3 > left) return false;
        const uint32_t seg = (regen + 3u) / 4u;
        if (3u * seg > regen) return false;
        sofs[0] = 6u; slen[0] = l1; sofs[1] = 6u + l1; slen[1] = l2; sofs[2] = 6u + l1 + l2; slen[2] = l3;
        sofs[3] = 6u + l1 + l2 + l3; slen[3] = left - sofs[3];
        cnt[0] = cnt[1] = cnt[2] = seg; cnt[3] = regen - 3u * seg;
        oofs[1] = seg; oofs[2] = 2u * seg; oofs[3] = 3u * seg;
    }
    const uint32_t nstreams = single ? 1u : 4u;
    const uint32_t me = lane < nstreams ? lane : 0u;
    const uint32_t my_ofs = me == 0u ? sofs[0] : (me == 1u ? sofs[1] : (me == 2u ? sofs[2] : sofs[3]));
    const uint32_t my_len = me == 0u ? slen[0] : (me == 1u ? slen[1] : (me == 2u ? slen[2] : slen[3]));
    const uint32_t my_cnt = me == 0u ? cnt[0] : (me == 1u ? cnt[1] : (me == 2u ? cnt[2] : cnt[3]));
    const uint32_t my_out = me == 0u ? oofs[0] : (me == 1u ? oofs[1] : (me == 2u ? oofs[2] : oofs[3]));
    bool ok = true;
    if (lane < nstreams) {
        BitRd b;
        ok = b.init(p + my_ofs, my_len);
        if (ok) {
            uint8_t *o = lit + my_out;
            for (uint32_t i = 0; i < my_cnt; i++) {
                const uint32_t e = L.huf[b.peek((uint32_t)hlog)];
                o[i] = (uint8_t)e;
                b.skip(e >> 8);
            }
            ok = (b.pos == 0) && !b.over; /* must end exactly */
        }
    }
    return wave_ballot(!ok) == 0ull;
}

struct FrameState {
    int huf_log;
    bool huf_valid, fse_valid;
    int ll_log, of_log, ml_log;
    uint32_t rep0, rep1, rep2;
};

/* one compressed block; returns false on malformed input */
__device__ bool decode_block(ZLds &L, Wave<ZR> &w, FrameState &fs, const uint8_t *src, uint32_t n, uint8_t *litbuf,
                             uint32_t cap, uint32_t lane, Stats &st)
{
    stamp(st, 7);
    /* ---------------- literals section ---------------- */
    if (n < 3u) return false;
    const uint32_t b0 = uni(src[0]);
    const uint32_t type = b0 & 3u, fmt = (b0 >> 2) & 3u;
    uint32_t regen, used;
    int lit_mode; /* 0 stream at lit_ptr, 1 RLE byte */
    const uint8_t *lit_ptr = nullptr;
    uint32_t rle_byte = 0;
    if (type < 2u) {
        uint32_t hdr;
        if (fmt == 1u) { hdr = 2; regen = (b0 >> 4) | (uni(src[1]) << 4); }
        else if (fmt == 3u) { hdr = 3; regen = (b0 >> 4) | (uni(src[1]) << 4) | (uni(src[2]) << 12); }
        else { hdr = 1; regen = b0 >> 3; }
        if (type == 0u) {
            if (hdr + regen > n || regen > kZBlockMax) return false;
            lit_mode = 0; lit_ptr = src + hdr; used = hdr + regen;
        } else {
            if ((fmt == 3u && n < 4u) || regen > kZBlockMax || hdr + 1u > n) return false;
            lit_mode = 1; rle_byte = uni(src[hdr]); used = hdr + 1u;
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
        if (type == 3u) { if (!fs.huf_valid) return false; }
        else {
            const int t = huf_read_table(L, L.huf, p, p, left, &fs.huf_log, lane);
            if (t < 0) return false;
            fs.huf_valid = true;
            p += t; left -= (uint32_t)t;
        }
        stamp(st, 0); /* literal header + huffman table */
        if (!huf_decode_streams(L, fs.huf_log, litbuf, regen, p, left, single, lane)) return false;
        stamp(st, 1); /* huffman streams */
        lit_mode = 0; lit_ptr = litbuf; used = hdr + csize;
    }
    /* make the decoded literals visible to the staging loads (same wave, in-order memory ops) */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    /* ---------------- sequences section ---------------- */
    const uint8_t *ip = src + used;
    uint32_t left = n - used;
    if (left < 1u) return false;
    uint32_t nseq = uni(ip[0]);
    ip++; left--;
    uint32_t lit_pos = 0; /* literals consumed */
    uint32_t lvp = 0;
    if (lit_mode == 0) lvp = stream_open(w, lit_ptr, regen);

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
        int u = read_seq_table(L, L.ll, &fs.ll_log, 0, (int)(modes >> 6), ip, left, fs.fse_valid);
        if (u < 0) return false;
        ip += u; left -= (uint32_t)u;
        u = read_seq_table(L, L.of, &fs.of_log, 1, (int)((modes >> 4) & 3u), ip, left, fs.fse_valid);
        if (u < 0) return false;
        ip += u; left -= (uint32_t)u;
        u = read_seq_table(L, L.ml, &fs.ml_log, 2, (int)((modes >> 2) & 3u), ip, left, fs.fse_valid);
        if (u < 0) return false;
        ip += u; left -= (uint32_t)u;
        fs.fse_valid = true;
        __builtin_amdgcn_wave_barrier();

        BitRd b;
        if (!b.init(ip, left)) return false;
        uint32_t sl = uni(b.read((uint32_t)fs.ll_log));
        uint32_t so = uni(b.read((uint32_t)fs.of_log));
        uint32_t sm = uni(b.read((uint32_t)fs.ml_log));
        /* Decoded sequences wait in a 64-entry queue held one-per-lane (q_ll, q_ml, q_off); its head
         * prefix of "simple" sequences is executed as ONE batch by the shared copy engine. */
        uint32_t q_ll = 0, q_ml = 0, q_off = 0;
        uint32_t qn = 0, decoded = 0;
        stamp(st, 2); /* sequence tables */
        for (;;) {
            stamp(st, 5);
            /* ---- refill the queue: serial FSE decode (one adaptive bitstream) ---- */
            while (qn < 64u && decoded < nseq) {
                const uint32_t el = L.ll[sl], eo = L.of[so], em = L.ml[sm];
                const uint32_t lsym = uni(el >> 14), osym = uni(eo >> 14), msym = uni(em >> 14);
                const uint32_t llbase = kLLBase[lsym], llbits = kLLBits[lsym];
                const uint32_t mlbase = kMLBase[msym], mlbits = kMLBits[msym];
                const bool ll0 = (llbase == 0u);
                uint32_t offset;
                if (osym > 1u) {
                    offset = ((1u << osym) - 3u) + uni(b.read(osym));
                    fs.rep2 = fs.rep1; fs.rep1 = fs.rep0; fs.rep0 = offset;
                } else if (osym == 0u) {
                    if (!ll0) offset = fs.rep0;
                    else { offset = fs.rep1; fs.rep1 = fs.rep0; fs.rep0 = offset; }
                } else {
                    const uint32_t idx = 1u + (ll0 ? 1u : 0u) + uni(b.read(1u));
                    uint32_t tmp = (idx == 3u) ? fs.rep0 - 1u : (idx == 1u ? fs.rep1 : fs.rep2);
                    if (tmp == 0u) tmp = 1u; /* 0 is not valid: forced to 1 like the library */
                    if (idx != 1u) fs.rep2 = fs.rep1;
                    fs.rep1 = fs.rep0;
                    fs.rep0 = offset = tmp;
                }
                const uint32_t mlen = mlbase + (mlbits ? uni(b.read(mlbits)) : 0u);
                const uint32_t llen = llbase + (llbits ? uni(b.read(llbits)) : 0u);
                decoded++;
                if (decoded < nseq) { /* state updates: LL, ML, OF */
                    sl = uni((el & 1023u) + b.read((el >> 10) & 15u));
                    sm = uni((em & 1023u) + b.read((em >> 10) & 15u));
                    so = uni((eo & 1023u) + b.read((eo >> 10) & 15u));
                }
                if (lane == qn) { q_ll = llen; q_ml = mlen; q_off = offset; }
                qn++;
            }
            if (qn == 0u) break;
            stamp(st, 3); /* FSE sequence decode */

            /* ---- head prefix of the queue that the batch engine can take ---- */
            const bool inq = lane < qn;
            const uint32_t outlen = inq ? q_ll + q_ml : 0u;
            const uint32_t oend = scan64_incl(outlen);
            const uint32_t ostart = oend - outlen;
            const uint32_t litend = scan64_incl(inq ? q_ll : 0u); /* literals consumed up to and incl. this sequence */
            const uint32_t mabs = w.op + ostart + q_ll;
            const bool isfar = inq && q_off >= ZR - kTMax;
            const bool ok = inq && lit_mode == 0 && (q_ml <= q_off || (q_off != 0u && q_ml <= 64u)) /* short self-overlap: a dependent match, lz4_copy.h */ && q_off <= mabs && !(isfar && q_ml > 32u) &&
                            litend <= regen - lit_pos && oend <= kTMax && (uint64_t)w.op + oend <= cap;
            const unsigned long long badmask = wave_ballot(!ok);
            const uint32_t nb = badmask ? ctz64(badmask) : 64u;
            if (nb > 0u) {
                const uint32_t T = lane_get(oend, nb - 1u);
                const uint32_t lits = lane_get(litend, nb - 1u);
                /* stage the literals of the whole batch (<= kTMax bytes) in the input ring */
                while (w.in_hi < w.vend && w.in_hi < lvp + lits + 8u) w.refill();
                asm volatile("" ::: "memory");
                if (lane < 2u) *reinterpret_cast<uint2 *>(L.in + kInRing + lane * 8u) = *reinterpret_cast<const uint2 *>(L.in + lane * 8u);
                else if (lane < 4u) *reinterpret_cast<uint2 *>(L.ring + ZR + (lane - 2u) * 8u) = *reinterpret_cast<const uint2 *>(L.ring + (lane - 2u) * 8u);
                w.flush();
                uint4 xfa = make_uint4(0, 0, 0, 0), xfb = xfa;
                if (lane < nb && isfar) {
                    const uint8_t *g = w.dst + (mabs - q_off);
                    __builtin_memcpy(&xfa, g, 16);
                    __builtin_memcpy(&xfb, g + 16, 16);
                }
                const CopyLds<ZR, kTMax> SL = {L.ring, L.in, L.meta, L.bm};
                seq_copy<ZR, kTMax>(w, SL, nb, ostart, q_ll, q_ml, q_off, lvp + (litend - q_ll), T, isfar, xfa, xfb, st);
                w.flush();
                lvp += lits;
                lit_pos += lits;
                /* drop the executed prefix from the queue */
                q_ll = __shfl(q_ll, (int)((lane + nb) & 63u), 64);
                q_ml = __shfl(q_ml, (int)((lane + nb) & 63u), 64);
                q_off = __shfl(q_off, (int)((lane + nb) & 63u), 64);
                qn -= nb;
            } else {
                /* the head sequence is not batchable (overlapping or very long match, RLE literals,
                 * or malformed): execute it alone, with every check */
                const uint32_t llen = lane_get(q_ll, 0), mlen = lane_get(q_ml, 0), offset = lane_get(q_off, 0);
                if (llen > regen - lit_pos) return false;
                if ((uint64_t)llen + mlen > (uint64_t)(cap - w.op)) return false;
                if (lit_mode == 0) lvp = wave_copy_literals(w, lvp, llen);
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
        if (b.over || b.pos != 0) return false; /* the bitstream must be consumed exactly */
    }
    /* last literals */
    const uint32_t rest = regen - lit_pos;
    if (rest > cap - w.op) return false;
    if (lit_mode == 0) wave_copy_literals(w, lvp, rest);
    else wave_fill(w, rle_byte, rest);
    w.flush();
    return true;
}

} // namespace

__global__ void __launch_bounds__(64)
k_zstd_dec(const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
           const uint32_t *__restrict__ src_size, uint8_t *dst_base, uint64_t dst_stride, uint32_t B,
           uint64_t n_blocks, int32_t *__restrict__ status, uint8_t *workspace, unsigned long long *stats,
           const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n, uint64_t list_base)
{
    __shared__ __attribute__((aligned(16))) ZLds L;
    const uint32_t lane = threadIdx.x & 63u;
    Stats st = {};
    st.on = stats != nullptr; /* diagnostic phase stamps (CRYO_ZSTD_STATS) */
    if (st.on) st.t0 = __builtin_amdgcn_s_memtime();
    uint8_t *litbuf = workspace + (uint64_t)blockIdx.x * kLitBuf;

    /* with a list (the batch pipeline's irregular frames) decode blocks list_base + list[0 .. *list_n) */
    if (list) n_blocks = uni(*list_n);
    for (uint64_t it = blockIdx.x; it < n_blocks; it += gridDim.x) {
        const uint64_t blk = list ? list_base + uni(list[it]) : it;
        const uint8_t *src = src_base + uni64(src_off[blk]);
        const uint32_t csize = uni(src_size[blk]);
        Wave<ZR> w;
        w.ring = L.ring;
        w.in = L.in;
        w.lane = lane;
        w.dst = dst_base + uni64(blk * dst_stride);
        w.dst_aligned = (reinterpret_cast<uintptr_t>(w.dst) & 15u) == 0;
        w.op = 0;
        w.flushed = 0;
        w.delta = 0; w.abase = src; w.vend = 0; w.in_hi = 0; w.pre = make_uint2(0, 0); w.pre2 = make_uint2(0, 0); w.pre3 = make_uint2(0, 0); w.nstale = 0;

        bool bad = false;
        uint32_t ip = 0;
        while (!bad && csize - ip >= 5u) {
            uint32_t magic;
            __builtin_memcpy(&magic, src + ip, 4);
            magic = uni(magic);
            if (csize - ip >= 8u && (magic & 0xFFFFFFF0u) == 0x184D2A50u) { /* skippable frame */
                uint32_t sz;
                __builtin_memcpy(&sz, src + ip + 4, 4);
                sz = uni(sz);
                if ((uint64_t)sz + 8u > csize - ip) { bad = true; break; }
                ip += 8u + sz;
                continue;
            }
            if (magic != 0xFD2FB528u) { bad = true; break; }
            const uint32_t fhd = uni(src[ip + 4]);
            const uint32_t single = (fhd >> 5) & 1u, did = fhd & 3u, fcs_flag = fhd >> 6, has_ck = (fhd >> 2) & 1u;
            const uint32_t did_sz = did == 3u ? 4u : did;
            const uint32_t fcs_sz = fcs_flag == 0u ? single : (1u << fcs_flag);
            const uint32_t hsz = 5u + (single ? 0u : 1u) + did_sz + fcs_sz;
            if ((fhd & 0x08u) || csize - ip < hsz) { bad = true; break; }
            uint32_t p = ip + 5u;
            if (!single) { if ((uni(src[p]) >> 3) + 10u > 31u) { bad = true; break; } p++; }
            if (did) {
                uint32_t id = 0;
                for (uint32_t k = 0; k < did_sz; k++) id |= uni(src[p + k]) << (8u * k);
                if (id != 0u) { bad = true; break; }
                p += did_sz;
            }
            uint64_t fcs = ~0ull;
            if (fcs_flag == 0u) { if (single) fcs = uni(src[p]); }
            else {
                uint64_t v = 0;
                for (uint32_t k = 0; k < fcs_sz; k++) v |= (uint64_t)uni(src[p + k]) << (8u * k);
                fcs = fcs_flag == 1u ? v + 256u : v;
            }
            ip += hsz;
            FrameState fs;
            fs.huf_log = 0; fs.huf_valid = false; fs.fse_valid = false;
            fs.ll_log = fs.of_log = fs.ml_log = 0;
            fs.rep0 = 1; fs.rep1 = 4; fs.rep2 = 8;
            const uint32_t frame_start = w.op;
            for (;;) {
                if (csize - ip < 3u) { bad = true; break; }
                const uint32_t bh = uni((uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16));
                ip += 3u;
                const uint32_t last = bh & 1u, type = (bh >> 1) & 3u, bsize = bh >> 3;
                if (type == 3u) { bad = true; break; }
                if (type == 1u) {
                    if (csize - ip < 1u || bsize > B - w.op) { bad = true; break; }
                    wave_fill(w, uni(src[ip]), bsize);
                    w.flush();
                    ip += 1u;
                } else {
                    if (bsize > csize - ip) { bad = true; break; }
                    if (type == 0u) {
                        if (bsize > B - w.op) { bad = true; break; }
                        const uint32_t vp = stream_open(w, src + ip, bsize);
                        wave_copy_literals(w, vp, bsize);
                        w.flush();
                    } else {
                        if (bsize >= kZBlockMax) { bad = true; break; }
                        if (!decode_block(L, w, fs, src + ip, bsize, litbuf, B, lane, st)) { bad = true; break; }
                    }
                    ip += bsize;
                }
                if (last) break;
            }
            if (bad) break;
            if (fcs != ~0ull && (uint64_t)(w.op - frame_start) != fcs) { bad = true; break; }
            if (has_ck) {
                if (csize - ip < 4u) { bad = true; break; }
                w.flush();
                w.flush_tail();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                uint32_t want;
                __builtin_memcpy(&want, src + ip, 4);
                if ((uint32_t)xxh64_dev(w.dst + frame_start, w.op - frame_start) != uni(want)) { bad = true; break; }
                ip += 4u;
            }
        }
        if (!bad && ip != csize) bad = true;
        if (!bad && w.op != B) bad = true;
        if (!bad) { w.flush(); w.flush_tail(); }
        if (lane == 0) status[blk] = bad ? CRYO_ST_CORRUPT : CRYO_ST_OK;
        __builtin_amdgcn_wave_barrier();
    }
    if (st.on && lane == 0) {
        stamp(st, 7);
        for (int k = 0; k < 8; k++) atomicAdd(&stats[k], st.t[k]);
    }
}

static uint32_t zstd_grid(uint64_t n_blocks)
{
    const uint64_t cap = 256u * 7u; /* 256 CUs x 7 workgroups (LDS bound) */
    return (uint32_t)(n_blocks < cap ? n_blocks : cap);
}

size_t zstd_fused_workspace(uint64_t n_blocks)
{
    return (size_t)zstd_grid(n_blocks) * kLitBuf + 256;
}

hipError_t launch_zstd_fused(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                             const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                             uint32_t block_size, uint64_t n_blocks, int32_t *d_status,
                             void *d_workspace, size_t workspace_bytes, const uint32_t *d_list,
                             const uint32_t *d_list_n, uint64_t list_base)
{
    if (n_blocks == 0) return hipSuccess;
    const uint32_t grid = zstd_grid(n_blocks);
    if (workspace_bytes < (size_t)grid * kLitBuf) return hipErrorInvalidValue;
    static const bool want_stats = cryo_tuning_env("CRYO_ZSTD_STATS") != nullptr; /* debugging aid */
    unsigned long long *d_st = nullptr, h_st[8];
    if (want_stats) {
        if (hipMalloc((void **)&d_st, sizeof h_st) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d_st, 0, sizeof h_st, s);
    }
    hipLaunchKernelGGL(k_zstd_dec, dim3(grid), dim3(64), 0, s, d_src, d_src_off, d_src_size, d_dst, dst_stride,
                       block_size, n_blocks, d_status, (uint8_t *)d_workspace, d_st, d_list, d_list_n, list_base);
    if (want_stats) {
        (void)hipMemcpyAsync(h_st, d_st, sizeof h_st, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        (void)hipFree(d_st);
        unsigned long long tot = 0;
        for (int k = 0; k < 8; k++) tot += h_st[k];
        static const char *nm[8] = {"lit hdr+huf table", "huffman streams", "seq tables", "FSE seq decode", "batch passA",
                                    "batch setup/general", "batch passB+flush", "frame/other"};
        for (int k = 0; k < 8; k++) fprintf(stderr, "[zstd cycles] %-20s %5.1f%%\n", nm[k], 100.0 * (double)h_st[k] / (double)(tot ? tot : 1));
    }
    return hipGetLastError();
}

} // namespace cryo
