/*
 * zstd_common.h -- device-side pieces shared by the Zstandard decode kernels (zstd_dec.hip: fused
 * one-wave-per-frame decoder; zstd_pipe.hip: four-stage batch pipeline).  Bit reader, FSE / Huffman
 * table construction (RFC 8878 4.1, 4.2.1), XXH64, and the glue to the shared LZ copy engine.
 * Table scratch lives in a caller-provided LDS struct `LT` with members
 *   huf[1<<12] (u16), norm[256] (i16), nxt[256] (u16), wdt[64] (u32), wts[256] (u8), cell[512] (u8).
 */
#pragma once
#include "lz_common.h"

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
template <class LT>
__device__ bool build_seq_table(LT &L, uint32_t *t, int max_sym, int log)
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
template <class LT>
__device__ int read_seq_table(LT &L, uint32_t *t, int *log, int kind, int mode, const uint8_t *src,
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
template <class LT>
__device__ int fse_decode_weights(LT &L, const uint8_t *src, uint32_t n)
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
template <class LT>
__device__ int huf_read_table(LT &L, const uint8_t *src, uint32_t n, int *hlog, uint32_t lane)
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

} // namespace
} // namespace cryo
