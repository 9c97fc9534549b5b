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

/* ---- per-lane backward bit reader fed through a private LDS ring ----
 * For kernels that run one bitstream PER LANE (zstd_pipe.hip: Huffman streams, FSE sequences).  A lane
 * that refills its container straight from global memory stalls the whole wave (s_waitcnt counts are per
 * wave, and the lanes refill at different times), so here the global loads happen at fixed program points
 * (`tick<J>`, J = static slot 0..3, one 16-byte block per lane per tick at most, consumed four ticks
 * later), land in a 128-byte per-lane ring in LDS (rows 0..31 of a [36][NL] dword array: mirror of the
 * address bits 0..6, transposed so lane l only touches bank l; rows 32..35 take the idle slots' stores), and the container is rebuilt from the ring with three aligned ds_reads.
 *
 * Contract: at most 12 bytes consumed between two ticks, at most 57 bits between two fills.  With one
 * block issued per tick while less than 112 bytes are reserved below the read position, at least 36
 * staged bytes are always ahead of the reader (zstd decode).  Bits below the stream start read
 * as zero, as in BitRd.  NL = lanes sharing the ring array.
 * Loads are whole aligned 16-byte blocks: up to 15 bytes after the stream end and down to the frame's
 * first byte rounded down to 16 are touched. */
template <int NL>
struct LaneBits {
    uint32_t *ring;        /* this lane's column: dword d is ring[d * NL] */
    const uint8_t *origin; /* the frame's first byte rounded down to 16; every position below is an offset from it
                              (a pointer rebuilt from an integer would turn the loads into FLAT accesses) */
    int32_t o_start, o_top; /* stream byte 0; 16-aligned end: block b = [o_top - 16(b+1), o_top - 16b) */
    int32_t pos;            /* unread bits */
    uint32_t issued, written, pend;
    bool alive;
    uint64_t c; /* container, left aligned */
    uint4 t0, t1, t2, t3;

    __device__ inline uint4 load_block(uint32_t idx) const
    {
        const int32_t a = o_top - 16 * (int32_t)(idx + 1u);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (a >= 0) v = *reinterpret_cast<const uint4 *>(origin + a);
        return v;
    }
    __device__ inline void write_block(uint4 v, uint32_t idx)
    {
        const int32_t a = o_top - 16 * (int32_t)(idx + 1u);
        if (a < o_start) { /* zero the bytes below the stream start */
            const int32_t k = o_start - a > 16 ? 16 : o_start - a;
            auto m = [&](int32_t q) { const int32_t nz = k - 4 * q; return nz >= 4 ? 0u : (nz <= 0 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * nz)); };
            v.x &= m(0); v.y &= m(1); v.z &= m(2); v.w &= m(3);
        }
        const uint32_t d0 = (uint32_t)(a >> 2) & 31u;
        ring[(d0 + 0u) * NL] = v.x;
        ring[(d0 + 1u) * NL] = v.y;
        ring[(d0 + 2u) * NL] = v.z;
        ring[(d0 + 3u) * NL] = v.w;
    }
    /* open the stream frame_src[off .. off+len) (false: empty or no end mark); primes the ring with its top 112 bytes */
    __device__ inline bool init(uint32_t *ring_col, const uint8_t *frame_src, uint32_t off, uint32_t len, bool on)
    {
        ring = ring_col;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(frame_src) & 15u);
        origin = frame_src - mis;
        o_start = (int32_t)(off + mis);
        o_top = (int32_t)((off + mis + len + 15u) & ~15u);
        pos = 0; issued = written = pend = 0; c = 0;
        t0 = t1 = t2 = t3 = make_uint4(0, 0, 0, 0);
        alive = on && len >= 1u;
        uint32_t last = 0;
        if (alive) last = frame_src[off + len - 1u];
        if (last == 0u) alive = false;
        if (!alive) { o_start = 16; o_top = 16; } /* idle ticks re-read origin[0..16) */
        if (alive) {
            pos = (int32_t)(len - 1u) * 8 + (31 - __builtin_clz(last));
            const uint4 a = load_block(0), b = load_block(1), c4 = load_block(2), d = load_block(3);
            const uint4 e = load_block(4), f = load_block(5), g = load_block(6);
            write_block(a, 0); write_block(b, 1); write_block(c4, 2); write_block(d, 3);
            write_block(e, 4); write_block(f, 5); write_block(g, 6);
            issued = written = 7;
        }
        return alive;
    }
    template <int J>
    __device__ inline uint4 &slot() { if constexpr (J == 0) return t0; else if constexpr (J == 1) return t1; else if constexpr (J == 2) return t2; else return t3; }
    /* fixed-cadence ring maintenance: retire the block loaded four ticks ago, maybe start another.
     * Branch-free around the memory operations -- exactly one global load and four LDS stores per tick,
     * whether or not the lane needs them (an idle slot re-reads the previous block and lands in the trash
     * rows 32..35) -- so the compiler can count the loads in flight (s_waitcnt vmcnt(3)) instead of draining. */
    template <int J>
    __device__ inline void tick()
    {
        {
            const bool pending = (pend >> J) & 1u;
            uint4 v = slot<J>();
            const int32_t a = o_top - 16 * (int32_t)(written + 1u);
            /* a block that reaches below the stream's first byte: once per stream, so the masks sit behind a branch the
             * WAVE takes (left to the compiler they were 28 selects in every tick; +1.8 % on the zstd decode rate.
             * Rows 0..1 mirrored behind row 31 to save the wrap arithmetic of fill() were measured too and lost 4 %: the
             * two extra LDS stores per tick cost a lone wave more than six address instructions per fill,
             * profiles/r03_variants_ab.txt) */
            if (wave_any(pending & (a < o_start))) {
                if (a < o_start) { /* zero the bytes below the stream start */
                    const int32_t k = o_start - a > 16 ? 16 : o_start - a;
                    auto m = [&](int32_t q) { const int32_t nz = k - 4 * q; return nz >= 4 ? 0u : (nz <= 0 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * nz)); };
                    v.x &= m(0); v.y &= m(1); v.z &= m(2); v.w &= m(3);
                }
            }
            const uint32_t d0 = pending ? (uint32_t)(a >> 2) & 31u : 32u;
            ring[(d0 + 0u) * NL] = v.x;
            ring[(d0 + 1u) * NL] = v.y;
            ring[(d0 + 2u) * NL] = v.z;
            ring[(d0 + 3u) * NL] = v.w;
            written += pending ? 1u : 0u;
        }
        const int32_t t = o_start + (pos > 0 ? (pos - 1) >> 3 : 0);
        const int32_t low = o_top - 16 * (int32_t)issued;
        const bool want = alive && t - low < 112 && low > o_start - 8;
        const uint32_t b = want ? issued : (issued ? issued - 1u : 0u);
        int32_t off = o_top - 16 * (int32_t)(b + 1u);
        off = off < 0 ? 0 : off; /* blocks below the origin lie wholly below the stream start: masked to zero above */
        slot<J>() = *reinterpret_cast<const uint4 *>(origin + off);
        issued += want ? 1u : 0u;
        pend = (pend & ~(1u << J)) | ((want ? 1u : 0u) << J);
    }
    /* rebuild the container: >= 57 valid bits (or all that are left) */
    __device__ inline void fill()
    {
        c = 0;
        if (pos > 0) {
            const int32_t lowoff = o_start + ((pos - 1) >> 3) - 7;
            const uint32_t d0 = (uint32_t)(lowoff >> 2), s = (uint32_t)lowoff & 3u;
            const uint32_t w0 = ring[(d0 & 31u) * NL], w1 = ring[((d0 + 1u) & 31u) * NL], w2 = ring[((d0 + 2u) & 31u) * NL];
            const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, s), hi = __builtin_amdgcn_alignbyte(w2, w1, s);
            c = (((uint64_t)hi << 32) | lo) << (7u - ((uint32_t)(pos - 1) & 7u));
        }
    }
    __device__ inline uint32_t peek(uint32_t nb) const { return (uint32_t)((c >> 1) >> (63u - nb)); } /* nb 0..32 */
    __device__ inline uint32_t peek_nz(uint32_t nb) const { return (uint32_t)(c >> 32) >> (32u - nb); } /* nb 1..32 */
    __device__ inline void skip(uint32_t nb) { c <<= nb; pos -= (int32_t)nb; }
    __device__ inline uint32_t read(uint32_t nb) { const uint32_t v = peek(nb); skip(nb); return v; }
};

__device__ inline uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)v, o, 64); v = x > v ? x : v; }
    return uni(v);
}

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
/* src: the bytes as the header parser reads them (may be a staged LDS copy); gsrc: the same bytes in global
 * memory for the bit reader */
template <class LT>
__device__ int fse_decode_weights(LT &L, const uint8_t *src, const uint8_t *gsrc, uint32_t n)
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
    if (!b.init(gsrc + hdr, n - (uint32_t)hdr)) return -1;
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

/* Huffman tree description -> table (LDS or global; symbol | nbits << 8).  Returns bytes consumed or -1;
 * *hlog = table log. */
template <class LT>
__device__ int huf_read_table(LT &L, uint16_t *table, const uint8_t *src, const uint8_t *gsrc, uint32_t n, int *hlog, uint32_t lane)
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
        nw = fse_decode_weights(L, src + 1, gsrc + 1, h0);
        if (nw < 0) return -1;
    }
    __builtin_amdgcn_wave_barrier();
    /* weight statistics, 64 weights at a time: how many symbols of every weight */
    uint32_t rank[kHufLogMax + 2];
#pragma unroll
    for (int r = 0; r < kHufLogMax + 2; r++) rank[r] = 0;
    for (int base = 0; base < nw; base += 64) {
        const int i = base + (int)lane;
        const bool on = i < nw;
        const uint32_t wv = on ? L.wts[i] : 0u;
        if (wave_any(on && wv >= (uint32_t)kHufLogMax)) return -1;
#pragma unroll
        for (int r = 1; r < kHufLogMax; r++) rank[r] += (uint32_t)__builtin_popcountll(wave_ballot(on && wv == (uint32_t)r));
    }
    uint32_t total = 0;
#pragma unroll
    for (int r = 1; r < kHufLogMax; r++) total += rank[r] << (r - 1);
    if (total == 0u) return -1;
    const int log = (int)hb32(total) + 1;
    if (log > kHufLogMax) return -1;
    const uint32_t rest = (1u << log) - total;
    if ((rest & (rest - 1u)) != 0u) return -1;
    const uint32_t lastw = hb32(rest) + 1u;
    if (lane == 0u) L.wts[nw] = (uint8_t)lastw;
#pragma unroll
    for (int r = 0; r < kHufLogMax + 1; r++) if (lastw == (uint32_t)r) rank[r]++;
    nw++;
    if (rank[1] < 2u || (rank[1] & 1u)) return -1;
    /* first cell and first place in the sorted symbol list of every weight class (cells: weight 1 first) */
    {
        uint32_t nx = 0, ns = 0;
#pragma unroll
        for (int r = 1; r <= kHufLogMax; r++) {
            if (lane == 0u) { L.wdt[r] = nx; L.wdt[16 + r] = ns; }
            nx += rank[r] << (r - 1);
            ns += rank[r];
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    /* symbols sorted by (weight, symbol): a symbol's place = its class's first place + the symbols of its weight before it */
    {
        uint32_t run[kHufLogMax + 1];
        {
            uint32_t ns = 0;
#pragma unroll
            for (int r = 1; r <= kHufLogMax; r++) { run[r] = ns; ns += rank[r]; }
        }
        for (int base = 0; base < nw; base += 64) {
            const int i = base + (int)lane;
            const bool on = i < nw;
            const uint32_t wv = on ? L.wts[i] : 0u;
            uint32_t place = 0;
#pragma unroll
            for (int r = 1; r <= kHufLogMax; r++) {
                const unsigned long long m = wave_ballot(on && wv == (uint32_t)r);
                if (wv == (uint32_t)r) place = run[r] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                run[r] += (uint32_t)__builtin_popcountll(m);
            }
            if (on && wv != 0u) L.cell[place] = (uint8_t)i;
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    /* canonical fill, two cells per lane: the cell's weight class from the class starts, the symbol from the sorted list */
    {
        uint32_t st[kHufLogMax + 1];
        {
            uint32_t nx = 0;
#pragma unroll
            for (int r = 1; r <= kHufLogMax; r++) { st[r] = nx; nx += rank[r] << (r - 1); }
        }
        uint32_t *t32 = reinterpret_cast<uint32_t *>(table);
        for (uint32_t u0 = 2u * lane; u0 < (1u << log); u0 += 128u) {
            uint32_t pair = 0;
#pragma unroll
            for (uint32_t h = 0; h < 2u; h++) {
                const uint32_t u = u0 + h;
                uint32_t cls = 1; /* the last non-empty class that starts at or below u */
#pragma unroll
                for (int q = 1; q <= kHufLogMax; q++) cls = (rank[q] != 0u && u >= st[q]) ? (uint32_t)q : cls;
                const uint32_t sym = L.cell[L.wdt[16 + cls] + ((u - L.wdt[cls]) >> (cls - 1u))];
                pair |= (sym | (((uint32_t)log + 1u - cls) << 8)) << (16u * h);
            }
            t32[u0 >> 1] = pair;
        }
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
    w.seg_on = false;
    w.delta = (uint32_t)(reinterpret_cast<uintptr_t>(ptr) & 15u);
    w.abase = ptr - w.delta;
    w.vend = w.delta + size;
    w.in_hi = 0;
    w.prefetch();
    if (size) { w.refill(); if (w.in_hi < w.vend) w.refill(); }
    return w.delta; /* virtual position of the first byte */
}

/* ... at a stream that lies in pieces (Wave::fetch_seg): lane j's piece = (offset into base, bytes, stream position) */
template <uint32_t R>
__device__ inline uint32_t stream_open_seg(Wave<R> &w, const uint8_t *base, uint4 piece, uint32_t size)
{
    w.seg_on = true;
    w.sbase = base;
    w.s_src = piece.x; w.s_dst = piece.z; w.s_end = piece.z + piece.y;
    {
        /* the next piece that is not empty (for the 8 bytes that straddle a piece's end) */
        const unsigned long long ne = __builtin_amdgcn_ballot_w64(piece.y != 0u);
        const unsigned long long later = w.lane >= 63u ? 0ull : (ne & (~0ull << (w.lane + 1u)));
        const uint32_t nl = later ? (uint32_t)__builtin_ctzll(later) : w.lane;
        w.s_nxt = (uint32_t)__shfl((int)piece.x, (int)nl, 64);
    }
    w.cs = 0;
    w.delta = 0;
    w.abase = base;
    w.vend = size;
    w.in_hi = 0;
    w.prefetch();
    if (size) { w.refill(); if (w.in_hi < w.vend) w.refill(); }
    return 0;
}

template <uint32_t R>
__device__ inline void wave_fill(Wave<R> &w, uint32_t byte, uint32_t len)
{
    uint32_t rem = len;
    while (rem) {
        w.flush();
        if (len - rem >= 16u && wave_stream_pattern(w, rem, false)) continue; /* once 16 bytes of the run are out */
        const uint32_t n = rem < 64u ? rem : 64u;
        if (w.lane < n) w.ring[(w.op + w.lane) & (R - 1)] = (uint8_t)byte;
        w.op += n;
        rem -= n;
    }
}

} // namespace
} // namespace cryo
