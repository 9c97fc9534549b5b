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
#include "lz_common.h"
#include <cstdio>
#include <cstdlib>

namespace cryo {

namespace {

constexpr uint32_t ZR = 4096;             /* output ring */
constexpr uint32_t kZBlockMax = 128u << 10;
constexpr uint32_t kLitBuf = kZBlockMax + 64u; /* per-workgroup literal buffer in global memory */
constexpr int kHufLogMax = 12;

/* ---- backward bit reader over global memory (per lane) ----
 * `cont` holds stream bits [cbase, cbase+64); `ahead` is the prefetched 64 bits just below it, so a
 * refill is a register funnel shift plus ONE new load that is not waited for until the next refill. */
struct BitRd {
    const uint8_t *p;
    uint32_t n;
    int32_t pos;   /* unread bits */
    int32_t cbase; /* bit index of cont bit 0 (multiple of 8) */
    uint64_t cont, ahead; /* ahead = bits [cbase-64, cbase) (zero below the stream start) */
    bool over;

    __device__ inline uint64_t load(int32_t bi) const /* 8 bytes at byte index bi (may be negative / past the end) */
    {
        uint64_t v = 0;
        if (bi >= 0 && (uint32_t)bi + 8u <= n) {
            __builtin_memcpy(&v, p + bi, 8);
        } else {
            for (int32_t k = 0; k < 8; k++) {
                const int32_t q = bi + k;
                if (q >= 0 && (uint32_t)q < n) v |= (uint64_t)p[q] << (8 * k);
            }
        }
        return v;
    }
    __device__ inline void refill()
    {
        int32_t nb = ((pos >> 3) - 7) * 8;
        if (nb < 0) nb = 0;
        const int32_t sh = cbase - nb; /* 0..64, multiple of 8 */
        if (sh >= 64) cont = ahead;
        else if (sh > 0) cont = (cont << sh) | (ahead >> (64 - sh));
        cbase = nb;
        ahead = load((nb >> 3) - 8);
    }
    __device__ inline bool init(const uint8_t *src, uint32_t len)
    {
        p = src; n = len; over = false; pos = 0; cbase = 0; cont = 0; ahead = 0;
        if (len < 1u) return false;
        const uint32_t last = src[len - 1u];
        if (last == 0u) return false;
        pos = (int32_t)(len - 1u) * 8 + (31 - __builtin_clz(last));
        int32_t bi = (pos >> 3) - 7;
        if (bi < 0) bi = 0;
        cbase = bi * 8;
        cont = load(bi);
        ahead = load(bi - 8);
        return true;
    }
    /* next nb (<= 32) bits, MSB first, zero extended past the start of the stream */
    __device__ inline uint32_t peek(uint32_t nb)
    {
        if (nb == 0u) return 0u;
        int32_t avail = pos - cbase;
        if (avail < (int32_t)nb && cbase > 0) { refill(); avail = pos - cbase; }
        if (avail >= (int32_t)nb) return (uint32_t)(cont >> (avail - (int32_t)nb)) & (uint32_t)((1ull << nb) - 1ull);
        if (avail <= 0) return 0u;
        return (uint32_t)((cont & ((1ull << avail) - 1ull)) << ((int32_t)nb - avail));
    }
    __device__ inline void skip(uint32_t nb) { pos -= (int32_t)nb; if (pos < 0) over = true; }
    __device__ inline uint32_t read(uint32_t nb) { const uint32_t v = peek(nb); skip(nb); return v; }
};

struct ZLds {
    uint8_t ring[ZR];
    uint8_t in[kInRing];
    uint16_t huf[1 << kHufLogMax]; /* symbol | nbits << 8 */
    uint32_t ll[512], ml[512], of[256]; /* next | nbits << 10 | symbol << 14 */
    int16_t norm[256];
    uint16_t nxt[256];
    uint32_t wdt[64]; /* FSE table of the Huffman weights */
    uint8_t wts[256];
    uint8_t cell[512];
    unsigned long long meta[64]; /* batch copy: per-sequence metadata */
    uint32_t bm[kTMax / 32];     /* batch copy: bitmap of sequence starts */
};

__device__ inline uint32_t hb32(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }

/* ---- FSE table description (forward, LSB-first bits); wave-uniform.  Returns bytes used or -1. */
__device__ int read_ncount(int16_t *norm, int *max_sym, int *table_log, const uint8_t *src, uint32_t n)
{
    uint64_t acc = 0;
    int nacc = 0, bits_used = 0;
    uint32_t fed = 0;
    const uint32_t nfeed = n < 4u ? 4u : n; /* the library pads short inputs to 4 bytes with zeros */
#define Z_FILL() while (nacc <= 56 && fed < nfeed) { acc |= (uint64_t)(fed < n ? uni(src[fed]) : 0u) << nacc; fed++; nacc += 8; }
#define Z_TAKE(k) do { acc >>= (k); nacc -= (k); bits_used += (k); } while (0)
    Z_FILL();
    int nb = (int)(acc & 15u) + 5;
    if (nb > 15) return -1;
    Z_TAKE(4);
    *table_log = nb;
    int remaining = (1 << nb) + 1, threshold = 1 << nb, sym = 0, prev0 = 0;
    nb++;
    while (remaining > 1 && sym <= *max_sym) {
        Z_FILL();
        if (prev0) {
            int n0 = sym;
            while ((acc & 0xFFFFu) == 0xFFFFu) { n0 += 24; Z_TAKE(16); Z_FILL(); if (nacc < 0) return -1; }
            while ((acc & 3u) == 3u) { n0 += 3; Z_TAKE(2); Z_FILL(); }
            n0 += (int)(acc & 3u);
            Z_TAKE(2);
            if (n0 > *max_sym) return -1;
            while (sym < n0) norm[sym++] = 0;
            Z_FILL();
        }
        const int max = (2 * threshold - 1) - remaining;
        int count;
        if ((int)(acc & (uint64_t)(threshold - 1)) < max) {
            count = (int)(acc & (uint64_t)(threshold - 1));
            Z_TAKE(nb - 1);
        } else {
            count = (int)(acc & (uint64_t)(2 * threshold - 1));
            if (count >= threshold) count -= max;
            Z_TAKE(nb);
        }
        count--;
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (int16_t)count;
        prev0 = !count;
        while (remaining < threshold) { nb--; threshold >>= 1; }
        if (nacc < 0) return -1;
    }
#undef Z_FILL
#undef Z_TAKE
    if (remaining != 1) return -1;
    *max_sym = sym - 1;
    const uint32_t used = (uint32_t)((bits_used + 7) >> 3);
    if (used > n) return -1;
    return (int)used;
}

/* spread symbols over 1<<log cells (wave-uniform) */
__device__ bool fse_spread(uint8_t *cell, uint16_t *nxt, const int16_t *norm, int max_sym, int log)
{
    const uint32_t size = 1u << log, mask = size - 1u, step = (size >> 1) + (size >> 3) + 3u;
    uint32_t high = size - 1u, pos = 0;
    for (int s = 0; s <= max_sym; s++) {
        const int c = norm[s];
        if (c == -1) { cell[high--] = (uint8_t)s; nxt[s] = 1; }
        else nxt[s] = (uint16_t)c;
    }
    for (int s = 0; s <= max_sym; s++) {
        const int c = norm[s];
        for (int i = 0; i < c; i++) {
            cell[pos] = (uint8_t)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    return pos == 0u;
}

__constant__ uint32_t kLLBase[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24,
    28, 32, 40, 48, 64, 0x80, 0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000};
__constant__ uint8_t kLLBits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6,
    7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ uint32_t kMLBase[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23,
    24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 0x83, 0x103, 0x203,
    0x403, 0x803, 0x1003, 0x2003, 0x4003, 0x8003, 0x10003};
__constant__ uint8_t kMLBits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ int16_t kLLDef[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3,
    2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ int16_t kMLDef[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__constant__ int16_t kOFDef[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1,
    -1, -1, -1};

/* build one sequence decoding table from normalized counts already in L.norm (wave-uniform) */
__device__ bool build_seq_table(ZLds &L, uint32_t *t, int max_sym, int log)
{
    if (!fse_spread(L.cell, L.nxt, L.norm, max_sym, log)) return false;
    const uint32_t size = 1u << log;
    for (uint32_t i = 0; i < size; i++) {
        const uint32_t s = L.cell[i];
        const uint32_t ns = L.nxt[s];
        L.nxt[s] = (uint16_t)(ns + 1u);
        const uint32_t nb = (uint32_t)log - hb32(ns);
        t[i] = ((ns << nb) - size) | (nb << 10) | (s << 14);
    }
    return true;
}

/* kind: 0 LL, 1 OF, 2 ML.  Returns bytes consumed or -1. */
__device__ int read_seq_table(ZLds &L, uint32_t *t, int *log, int kind, int mode, const uint8_t *src,
                              uint32_t n, bool have_prev)
{
    const int max_sym_k = kind == 0 ? 35 : (kind == 1 ? 31 : 52);
    const int max_log_k = kind == 1 ? 8 : 9;
    if (mode == 0) {
        const int16_t *def = kind == 0 ? kLLDef : (kind == 1 ? kOFDef : kMLDef);
        const int ms = kind == 0 ? 35 : (kind == 1 ? 28 : 52);
        for (int i = 0; i <= ms; i++) L.norm[i] = def[i];
        *log = kind == 1 ? 5 : 6;
        return build_seq_table(L, t, ms, *log) ? 0 : -1;
    }
    if (mode == 1) {
        if (n < 1u) return -1;
        const uint32_t s = uni(src[0]);
        if ((int)s > max_sym_k) return -1;
        *log = 0;
        t[0] = s << 14;
        return 1;
    }
    if (mode == 2) {
        int ms = max_sym_k, lg = 0;
        const int used = read_ncount(L.norm, &ms, &lg, src, n);
        if (used < 0 || lg > max_log_k) return -1;
        if (!build_seq_table(L, t, ms, lg)) return -1;
        *log = lg;
        return used;
    }
    return have_prev ? 0 : -1;
}

/* FSE-compressed Huffman weights (wave-uniform).  Returns number of weights or -1. */
__device__ int fse_decode_weights(ZLds &L, const uint8_t *src, uint32_t n)
{
    int max_sym = 255, log = 0;
    const int hdr = read_ncount(L.norm, &max_sym, &log, src, n);
    if (hdr < 0 || log > 6) return -1;
    if (!fse_spread(L.cell, L.nxt, L.norm, max_sym, log)) return -1;
    uint32_t *dt = L.wdt;
    const uint32_t size = 1u << log;
    for (uint32_t i = 0; i < size; i++) {
        const uint32_t s = L.cell[i];
        const uint32_t ns = L.nxt[s];
        L.nxt[s] = (uint16_t)(ns + 1u);
        const uint32_t nb = (uint32_t)log - hb32(ns);
        dt[i] = ((ns << nb) - size) | (nb << 10) | (s << 14);
    }
    BitRd b;
    if (!b.init(src + hdr, n - (uint32_t)hdr)) return -1;
    uint32_t s1 = uni(b.read((uint32_t)log));
    uint32_t s2 = uni(b.read((uint32_t)log));
    int out = 0;
    for (;;) {
        if (out > 255 - 2) return -1;
        uint32_t e = dt[s1];
        L.wts[out++] = (uint8_t)(e >> 14);
        s1 = (e & 1023u) + uni(b.read((e >> 10) & 15u));
        if (b.over) { L.wts[out++] = (uint8_t)(dt[s2] >> 14); break; }
        if (out > 255 - 2) return -1;
        e = dt[s2];
        L.wts[out++] = (uint8_t)(e >> 14);
        s2 = (e & 1023u) + uni(b.read((e >> 10) & 15u));
        if (b.over) { L.wts[out++] = (uint8_t)(dt[s1] >> 14); break; }
    }
    return out;
}

/* Huffman tree description -> L.huf.  Returns bytes consumed or -1; *hlog = table log. */
__device__ int huf_read_table(ZLds &L, const uint8_t *src, uint32_t n, int *hlog, uint32_t lane)
{
    if (n < 1u) return -1;
    const uint32_t h0 = uni(src[0]);
    int nw, used;
    if (h0 >= 128u) {
        nw = (int)h0 - 127;
        used = 1 + (nw + 1) / 2;
        if ((uint32_t)used > n) return -1;
        for (int i = (int)lane; i < nw; i += 64) {
            const uint32_t byte = src[1 + i / 2];
            L.wts[i] = (uint8_t)((i & 1) ? (byte & 15u) : (byte >> 4));
        }
    } else {
        used = 1 + (int)h0;
        if ((uint32_t)used > n) return -1;
        nw = fse_decode_weights(L, src + 1, h0);
        if (nw < 0) return -1;
    }
    __builtin_amdgcn_wave_barrier();
    /* weight statistics (wave-uniform over <= 255 weights) */
    uint32_t rank[kHufLogMax + 2];
#pragma unroll
    for (int r = 0; r < kHufLogMax + 2; r++) rank[r] = 0;
    uint32_t total = 0;
    for (int i = 0; i < nw; i++) {
        const uint32_t wv = uni(L.wts[i]);
        if (wv >= (uint32_t)kHufLogMax) return -1;
#pragma unroll
        for (int r = 0; r < kHufLogMax; r++) if (wv == (uint32_t)r) rank[r]++;
        total += (1u << wv) >> 1;
    }
    if (total == 0u) return -1;
    const int log = (int)hb32(total) + 1;
    if (log > kHufLogMax) return -1;
    const uint32_t rest = (1u << log) - total;
    if ((rest & (rest - 1u)) != 0u) return -1;
    const uint32_t lastw = hb32(rest) + 1u;
    L.wts[nw] = (uint8_t)lastw;
#pragma unroll
    for (int r = 0; r < kHufLogMax + 1; r++) if (lastw == (uint32_t)r) rank[r]++;
    nw++;
    if (rank[1] < 2u || (rank[1] & 1u)) return -1;
    uint32_t start[kHufLogMax + 2];
    {
        uint32_t nx = 0;
#pragma unroll
        for (int r = 1; r <= kHufLogMax; r++) { start[r] = nx; nx += rank[r] << (r - 1); }
        start[0] = 0; start[kHufLogMax + 1] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    /* canonical fill: symbols in increasing order inside a weight; lanes cover each symbol's cells */
    for (int i = 0; i < nw; i++) {
        const uint32_t wv = uni(L.wts[i]);
        if (wv == 0u) continue;
        uint32_t st = 0;
#pragma unroll
        for (int r = 1; r <= kHufLogMax; r++) if (wv == (uint32_t)r) { st = start[r]; start[r] += (1u << wv) >> 1; }
        const uint32_t len = (1u << wv) >> 1;
        const uint16_t ent = (uint16_t)((uint32_t)i | (((uint32_t)log + 1u - wv) << 8));
        for (uint32_t u = lane; u < len; u += 64u) L.huf[st + u] = ent;
    }
    __builtin_amdgcn_wave_barrier();
    *hlog = log;
    return used;
}

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
    return __ballot(!ok) == 0ull;
}

/* XXH64 of the decoded frame (content checksum); wave-uniform, rare path */
__device__ inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ uint64_t xxh64_dev(const uint8_t *p, uint32_t len)
{
    const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull,
                   P4 = 9650029242287828579ull, P5 = 2870177450012600261ull;
    auto rd64 = [&](uint32_t o) { uint64_t v; __builtin_memcpy(&v, p + o, 8); return uni64(v); };
    auto rd32 = [&](uint32_t o) { uint32_t v; __builtin_memcpy(&v, p + o, 4); return uni(v); };
    auto rnd = [&](uint64_t a, uint64_t v) { return rotl64(a + v * P2, 31) * P1; };
    auto mrg = [&](uint64_t h, uint64_t v) { return (h ^ rnd(0, v)) * P1 + P4; };
    uint32_t o = 0;
    uint64_t h;
    if (len >= 32u) {
        uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
        do { v1 = rnd(v1, rd64(o)); v2 = rnd(v2, rd64(o + 8)); v3 = rnd(v3, rd64(o + 16)); v4 = rnd(v4, rd64(o + 24)); o += 32u; }
        while (o + 32u <= len);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = mrg(h, v1); h = mrg(h, v2); h = mrg(h, v3); h = mrg(h, v4);
    } else h = P5;
    h += len;
    while (o + 8u <= len) { h ^= rnd(0, rd64(o)); h = rotl64(h, 27) * P1 + P4; o += 8u; }
    if (o + 4u <= len) { h ^= (uint64_t)rd32(o) * P1; h = rotl64(h, 23) * P2 + P3; o += 4u; }
    while (o < len) { h ^= (uint64_t)uni(p[o]) * P5; h = rotl64(h, 11) * P1; o++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

/* point the wave's input ring at a byte stream in global memory */
template <uint32_t R>
__device__ inline uint32_t stream_open(Wave<R> &w, const uint8_t *ptr, uint32_t size)
{
    w.delta = (uint32_t)(reinterpret_cast<uintptr_t>(ptr) & 15u);
    w.abase = ptr - w.delta;
    w.vend = w.delta + size;
    w.in_hi = 0;
    w.prefetch();
    if (size) { w.refill(); if (w.in_hi < w.vend) w.refill(); }
    return w.delta; /* virtual position of the first byte */
}

template <uint32_t R>
__device__ inline void wave_fill(Wave<R> &w, uint32_t byte, uint32_t len)
{
    uint32_t rem = len;
    while (rem) {
        w.flush();
        const uint32_t n = rem < 64u ? rem : 64u;
        if (w.lane < n) w.ring[(w.op + w.lane) & (R - 1)] = (uint8_t)byte;
        w.op += n;
        rem -= n;
    }
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
            const int t = huf_read_table(L, p, left, &fs.huf_log, lane);
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
            const bool ok = inq && lit_mode == 0 && q_ml <= q_off && q_off <= mabs && q_off < (1u << 21) &&
                            litend <= regen - lit_pos && oend <= kTMax && (uint64_t)w.op + oend <= cap;
            const unsigned long long badmask = __ballot(!ok);
            const uint32_t nb = badmask ? ctz64(badmask) : 64u;
            if (nb > 0u) {
                const uint32_t T = lane_get(oend, nb - 1u);
                const uint32_t lits = lane_get(litend, nb - 1u);
                /* stage the literals of the whole batch (<= kTMax bytes) in the input ring */
                while (w.in_hi < w.vend && w.in_hi < lvp + lits + 8u) w.refill();
                batch_copy<ZR>(w, L.in, L.meta, L.bm, nb, ostart, q_ll, q_off, (lvp + (litend - q_ll)) - ostart, T, st);
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
           uint64_t n_blocks, int32_t *__restrict__ status, uint8_t *workspace, unsigned long long *stats)
{
    __shared__ __attribute__((aligned(16))) ZLds L;
    const uint32_t lane = threadIdx.x & 63u;
    Stats st = {};
    st.on = stats != nullptr; /* diagnostic phase stamps (CRYO_ZSTD_STATS) */
    if (st.on) st.t0 = __builtin_amdgcn_s_memtime();
    uint8_t *litbuf = workspace + (uint64_t)blockIdx.x * kLitBuf;

    for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
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
        w.delta = 0; w.abase = src; w.vend = 0; w.in_hi = 0; w.pre = make_uint2(0, 0);

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

size_t zstd_decompress_workspace(uint64_t n_blocks, uint32_t)
{
    return (size_t)zstd_grid(n_blocks) * kLitBuf + 256;
}

hipError_t launch_zstd_decompress(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                                  const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                                  uint32_t block_size, uint64_t n_blocks, int32_t *d_status,
                                  void *d_workspace, size_t workspace_bytes)
{
    if (n_blocks == 0) return hipSuccess;
    const uint32_t grid = zstd_grid(n_blocks);
    if (workspace_bytes < (size_t)grid * kLitBuf) return hipErrorInvalidValue;
    static const bool want_stats = getenv("CRYO_ZSTD_STATS") != nullptr; /* debugging aid */
    unsigned long long *d_st = nullptr, h_st[8];
    if (want_stats) {
        if (hipMalloc((void **)&d_st, sizeof h_st) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d_st, 0, sizeof h_st, s);
    }
    hipLaunchKernelGGL(k_zstd_dec, dim3(grid), dim3(64), 0, s, d_src, d_src_off, d_src_size, d_dst, dst_stride,
                       block_size, n_blocks, d_status, (uint8_t *)d_workspace, d_st);
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
