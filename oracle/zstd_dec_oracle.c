/*
 * zstd_dec_oracle.c -- CPU ORACLE (test infrastructure, see cryo_oracle.h).
 *
 * Restates the Zstandard frame decoder for the reference's call shape
 *   compression.c:116  ZSTD_decompress(out, CRYO_BLCKSZ, compressed, compressed_size)
 * libzstd is a third-party dependency of the reference (Makefile:5 -lzstd), not vendored
 * and not version-pinned there.  The format is frozen (RFC 8878), so decoding does not depend
 * on the library version.  Malformed input: every structural check of libzstd 1.4.8's
 * one-shot decoder is restated; in addition every entropy bitstream (Huffman streams, FSE
 * weight stream excepted by design, sequence stream) must be consumed exactly, as RFC 8878
 * requires.  libzstd 1.4.x is laxer in two places whose results depend on reading past the
 * start of a bitstream (the double-symbol Huffman decoder clamps an over-read on the last
 * symbol; the sequence decoder does not test for over-read after the last sequence); such
 * blocks are rejected here.  Pinned by tests/golden/adversarial.json and a live fuzz against
 * libzstd.so.1:  oracle accepts => library accepts with identical bytes;
 *                library rejects => oracle rejects.
 *
 * Scope: everything ZSTD_compress() of any level can emit and everything the one-shot
 * decoder accepts without a dictionary: concatenated frames, skippable frames, raw / RLE /
 * compressed blocks, raw / RLE / Huffman (1 or 4 streams) / treeless literals, predefined /
 * RLE / FSE / repeat sequence tables, repeat offsets, optional content checksum (XXH64).
 */
#include "cryo_oracle.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#define ZBLOCK_MAX (128u * 1024u)
#define HUF_LOG_MAX 12
#define ERR (-1L)

typedef struct { uint16_t base_state; uint8_t nbits; uint8_t sym; } fse_ent; /* generic FSE */
typedef struct { uint32_t base; uint16_t next; uint8_t nbits; uint8_t extra; } seq_ent;

typedef struct {
    /* entropy state carried across the blocks of one frame */
    uint8_t huf_sym[1 << HUF_LOG_MAX], huf_nb[1 << HUF_LOG_MAX];
    int huf_log, huf_valid;
    seq_ent ll[512], of[256], ml[512];
    int ll_log, of_log, ml_log, fse_valid;
    uint32_t rep[3];
    uint8_t lit[ZBLOCK_MAX + 32];
} zctx;

static int hb32(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; } /* v != 0 */

/* ------------------------------------------------------------------ backward bit reader
 * Bits are numbered from the least significant bit of byte 0; `pos` counts the unread bits.
 * Reading past the beginning yields zero bits and sets `over` (libzstd: BIT_DStream_overflow). */
typedef struct { const uint8_t *p; int64_t pos; int over; } brd;

static int br_init(brd *b, const uint8_t *p, size_t n)
{
    if (n < 1 || p[n - 1] == 0) return -1;
    b->p = p;
    b->pos = (int64_t)(n - 1) * 8 + hb32(p[n - 1]); /* drop padding and the end mark */
    b->over = 0;
    return 0;
}
static uint32_t br_peek(const brd *b, int n) /* next n (<= 24) bits, MSB first, zero extended */
{
    uint32_t v = 0;
    int i;
    for (i = 0; i < n; i++) {
        int64_t q = b->pos - 1 - i;
        v <<= 1;
        if (q >= 0) v |= (b->p[q >> 3] >> (q & 7)) & 1u;
    }
    return v;
}
static void br_skip(brd *b, int n) { b->pos -= n; if (b->pos < 0) b->over = 1; }
static uint32_t br_read(brd *b, int n) { uint32_t v = br_peek(b, n); br_skip(b, n); return v; }

/* ------------------------------------------------------------------ FSE table description */
static long read_ncount(int16_t *norm, int *max_sym, int *table_log, const uint8_t *src, size_t n)
{
    uint8_t buf[8];
    const uint8_t *ip = src;
    uint64_t acc;   /* forward LSB-first bit accumulator, refilled bytewise */
    int nacc, nb, remaining, threshold, sym = 0, prev0 = 0, bits_used = 0;
    size_t fed = 0;
    if (n < 4) { /* the library pads short inputs to 4 bytes and re-runs */
        long r;
        memset(buf, 0, sizeof buf);
        memcpy(buf, src, n);
        r = read_ncount(norm, max_sym, table_log, buf, 4);
        if (r < 0 || (size_t)r > n) return ERR;
        return r;
    }
    acc = 0; nacc = 0;
#define FILL() while (nacc <= 56 && fed < n) { acc |= (uint64_t)ip[fed++] << nacc; nacc += 8; }
#define TAKE(k) do { acc >>= (k); nacc -= (k); bits_used += (k); } while (0)
    FILL();
    nb = (int)(acc & 15) + 5;
    if (nb > 15) return ERR;
    TAKE(4);
    *table_log = nb;
    remaining = (1 << nb) + 1;
    threshold = 1 << nb;
    nb++;
    while (remaining > 1 && sym <= *max_sym) {
        FILL();
        if (prev0) {
            int n0 = sym;
            while ((acc & 0xFFFF) == 0xFFFF) { n0 += 24; TAKE(16); FILL(); if (nacc < 0) return ERR; }
            while ((acc & 3) == 3) { n0 += 3; TAKE(2); FILL(); }
            n0 += (int)(acc & 3);
            TAKE(2);
            if (n0 > *max_sym) return ERR;
            while (sym < n0) norm[sym++] = 0;
            FILL();
        }
        {
            int max = (2 * threshold - 1) - remaining, count;
            if ((int)(acc & (uint64_t)(threshold - 1)) < max) {
                count = (int)(acc & (uint64_t)(threshold - 1));
                TAKE(nb - 1);
            } else {
                count = (int)(acc & (uint64_t)(2 * threshold - 1));
                if (count >= threshold) count -= max;
                TAKE(nb);
            }
            count--; /* -1 = probability "less than one" */
            remaining -= count < 0 ? -count : count;
            norm[sym++] = (int16_t)count;
            prev0 = !count;
            while (remaining < threshold) { nb--; threshold >>= 1; }
        }
        if (nacc < 0) return ERR; /* ran past the input */
    }
#undef FILL
#undef TAKE
    if (remaining != 1) return ERR;
    *max_sym = sym - 1;
    {
        size_t used = (size_t)((bits_used + 7) >> 3);
        if (used > n) return ERR;
        return (long)used;
    }
}

/* spread symbols over a table of size 1<<log; returns per-cell symbol, fills next[] */
static int fse_spread(uint8_t *cell_sym, uint16_t *next, const int16_t *norm, int max_sym, int log)
{
    const uint32_t size = 1u << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint32_t high = size - 1, pos = 0;
    int s, i;
    for (s = 0; s <= max_sym; s++) {
        if (norm[s] == -1) { cell_sym[high--] = (uint8_t)s; next[s] = 1; }
        else next[s] = (uint16_t)norm[s];
    }
    for (s = 0; s <= max_sym; s++) {
        for (i = 0; i < norm[s]; i++) {
            cell_sym[pos] = (uint8_t)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    return pos == 0 ? 0 : -1;
}

/* ------------------------------------------------------------------ Huffman */
static long fse_decode_weights(uint8_t *w, int cap, const uint8_t *src, size_t n)
{
    int16_t norm[256];
    uint8_t cell[64];
    uint16_t next[256];
    fse_ent dt[64];
    int max_sym = 255, log, i, out = 0;
    long hdr = read_ncount(norm, &max_sym, &log, src, n);
    brd b;
    uint32_t s1, s2;
    if (hdr < 0 || log > 6) return ERR;
    if (fse_spread(cell, next, norm, max_sym, log)) return ERR;
    for (i = 0; i < (1 << log); i++) {
        uint32_t ns = next[cell[i]]++;
        int nb = log - hb32(ns);
        dt[i].sym = cell[i];
        dt[i].nbits = (uint8_t)nb;
        dt[i].base_state = (uint16_t)((ns << nb) - (1u << log));
    }
    if (br_init(&b, src + hdr, n - (size_t)hdr)) return ERR;
    s1 = br_read(&b, log);
    s2 = br_read(&b, log);
    /* two interleaved states; the stream ends by running dry: the state whose update
     * overran is dropped, the other one still yields its symbol */
    for (;;) {
        if (out > cap - 2) return ERR;
        w[out++] = dt[s1].sym;
        s1 = dt[s1].base_state + br_read(&b, dt[s1].nbits);
        if (b.over) { w[out++] = dt[s2].sym; break; }
        if (out > cap - 2) return ERR;
        w[out++] = dt[s2].sym;
        s2 = dt[s2].base_state + br_read(&b, dt[s2].nbits);
        if (b.over) { w[out++] = dt[s1].sym; break; }
    }
    return out;
}

/* reads a Huffman tree description; returns bytes consumed */
static long huf_read_table(zctx *z, const uint8_t *src, size_t n)
{
    uint8_t w[256];
    uint32_t rank[HUF_LOG_MAX + 2], start[HUF_LOG_MAX + 2];
    uint32_t total = 0, rest;
    long nw, used;
    int i, log;
    if (n < 1) return ERR;
    if (src[0] >= 128) {
        nw = src[0] - 127;
        used = 1 + (nw + 1) / 2;
        if ((size_t)used > n) return ERR;
        for (i = 0; i < nw; i += 2) {
            w[i] = src[1 + i / 2] >> 4;
            if (i + 1 < 256) w[i + 1] = src[1 + i / 2] & 15;
        }
    } else {
        used = 1 + src[0];
        if ((size_t)used > n) return ERR;
        nw = fse_decode_weights(w, 255, src + 1, src[0]);
        if (nw < 0) return ERR;
    }
    memset(rank, 0, sizeof rank);
    for (i = 0; i < nw; i++) {
        if (w[i] >= HUF_LOG_MAX) return ERR;
        rank[w[i]]++;
        total += (1u << w[i]) >> 1;
    }
    if (total == 0) return ERR;
    log = hb32(total) + 1;
    if (log > HUF_LOG_MAX) return ERR;
    rest = (1u << log) - total;
    if ((rest & (rest - 1)) != 0) return ERR; /* the implied last weight must be a power of two */
    w[nw] = (uint8_t)(hb32(rest) + 1);
    rank[w[nw]]++;
    nw++;
    if (rank[1] < 2 || (rank[1] & 1)) return ERR;
    /* canonical table: weight 1 first, symbols in increasing order inside a weight */
    {
        uint32_t nxt = 0;
        int r;
        for (r = 1; r <= log; r++) { start[r] = nxt; nxt += rank[r] << (r - 1); }
    }
    for (i = 0; i < nw; i++) {
        if (w[i]) {
            uint32_t len = (1u << w[i]) >> 1, u;
            for (u = 0; u < len; u++) {
                z->huf_sym[start[w[i]] + u] = (uint8_t)i;
                z->huf_nb[start[w[i]] + u] = (uint8_t)(log + 1 - w[i]);
            }
            start[w[i]] += len;
        }
    }
    z->huf_log = log;
    z->huf_valid = 1;
    return used;
}

static int huf_stream(const zctx *z, uint8_t *dst, size_t count, const uint8_t *src, size_t n)
{
    brd b;
    size_t i;
    if (br_init(&b, src, n)) return -1;
    for (i = 0; i < count; i++) {
        uint32_t v = br_peek(&b, z->huf_log);
        dst[i] = z->huf_sym[v];
        br_skip(&b, z->huf_nb[v]);
    }
    return (b.pos == 0 && !b.over) ? 0 : -1; /* must end exactly */
}

/* ------------------------------------------------------------------ literals section */
static long decode_literals(zctx *z, const uint8_t *src, size_t n, size_t *lit_size)
{
    uint32_t type, fmt, regen, csize, hdr;
    if (n < 3) return ERR;
    type = src[0] & 3;
    fmt = (src[0] >> 2) & 3;
    if (type < 2) { /* raw / RLE */
        if (fmt == 1) { hdr = 2; regen = (src[0] >> 4) | ((uint32_t)src[1] << 4); }
        else if (fmt == 3) { hdr = 3; regen = (src[0] >> 4) | ((uint32_t)src[1] << 4) | ((uint32_t)src[2] << 12); }
        else { hdr = 1; regen = src[0] >> 3; }
        if (type == 0) {
            if (hdr + regen > n) return ERR;
            if (regen > ZBLOCK_MAX) return ERR;
            memcpy(z->lit, src + hdr, regen);
            *lit_size = regen;
            return (long)(hdr + regen);
        }
        if (fmt == 3 && n < 4) return ERR;
        if (regen > ZBLOCK_MAX) return ERR;
        if (hdr + 1 > n) return ERR;
        memset(z->lit, src[hdr], regen);
        *lit_size = regen;
        return (long)(hdr + 1);
    }
    /* Huffman compressed (2) / treeless (3) */
    {
        int single = 0;
        uint32_t h;
        if (n < 5) return ERR;
        h = (uint32_t)src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | ((uint32_t)src[3] << 24);
        if (fmt == 0 || fmt == 1) { single = !fmt; hdr = 3; regen = (h >> 4) & 0x3FF; csize = (h >> 14) & 0x3FF; }
        else if (fmt == 2) { hdr = 4; regen = (h >> 4) & 0x3FFF; csize = h >> 18; }
        else { hdr = 5; regen = (h >> 4) & 0x3FFFF; csize = (h >> 22) + ((uint32_t)src[4] << 10); }
        if (regen > ZBLOCK_MAX) return ERR;
        if (csize + hdr > n) return ERR;
        {
            const uint8_t *p = src + hdr;
            size_t left = csize;
            if (type == 3) { if (!z->huf_valid) return ERR; }
            else {
                long t = huf_read_table(z, p, left);
                if (t < 0) return ERR;
                p += t; left -= (size_t)t;
            }
            if (single) {
                if (huf_stream(z, z->lit, regen, p, left)) return ERR;
            } else {
                size_t l1, l2, l3, l4, seg = (regen + 3) / 4;
                if (left < 10) return ERR;
                l1 = p[0] | (p[1] << 8); l2 = p[2] | (p[3] << 8); l3 = p[4] | (p[5] << 8);
                if (6 + l1 + l2 + l3 > left) return ERR;
                l4 = left - 6 - l1 - l2 - l3;
                if (3 * seg > regen) return ERR;
                p += 6;
                if (huf_stream(z, z->lit, seg, p, l1)) return ERR;
                if (huf_stream(z, z->lit + seg, seg, p + l1, l2)) return ERR;
                if (huf_stream(z, z->lit + 2 * seg, seg, p + l1 + l2, l3)) return ERR;
                if (huf_stream(z, z->lit + 3 * seg, regen - 3 * seg, p + l1 + l2 + l3, l4)) return ERR;
            }
        }
        *lit_size = regen;
        return (long)(hdr + csize);
    }
}

/* ------------------------------------------------------------------ sequences */
static const uint32_t LL_base[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24,
    28, 32, 40, 48, 64, 0x80, 0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000};
static const uint8_t LL_bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6,
    7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t ML_base[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23,
    24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 0x83, 0x103, 0x203,
    0x403, 0x803, 0x1003, 0x2003, 0x4003, 0x8003, 0x10003};
static const uint8_t ML_bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const int16_t LL_def[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3,
    2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t ML_def[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t OF_def[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1,
    -1, -1, -1};

/* kind: 0 = LL, 1 = OF, 2 = ML */
static void sym_info(int kind, int s, uint32_t *base, uint8_t *extra)
{
    if (kind == 0) { *base = LL_base[s]; *extra = LL_bits[s]; }
    else if (kind == 2) { *base = ML_base[s]; *extra = ML_bits[s]; }
    else { *base = s < 2 ? (uint32_t)s : (1u << s) - 3u; *extra = (uint8_t)s; } /* OF_base: 0,1,1,5,0xD,... */
}

static int build_seq_table(seq_ent *t, int kind, const int16_t *norm, int max_sym, int log)
{
    uint8_t cell[512];
    uint16_t next[64];
    int i;
    if (fse_spread(cell, next, norm, max_sym, log)) return -1;
    for (i = 0; i < (1 << log); i++) {
        uint32_t ns = next[cell[i]]++;
        int nb = log - hb32(ns);
        t[i].nbits = (uint8_t)nb;
        t[i].next = (uint16_t)((ns << nb) - (1u << log));
        sym_info(kind, cell[i], &t[i].base, &t[i].extra);
    }
    return 0;
}

/* one of the three table descriptions; returns bytes consumed */
static long read_seq_table(seq_ent *t, int *log, int kind, int mode, const uint8_t *src, size_t n, int have_prev)
{
    static const int max_sym_k[3] = {35, 31, 52}, max_log_k[3] = {9, 8, 9};
    int16_t norm[64];
    if (mode == 0) {
        if (kind == 0) { *log = 6; return build_seq_table(t, 0, LL_def, 35, 6) ? ERR : 0; }
        if (kind == 1) { *log = 5; return build_seq_table(t, 1, OF_def, 28, 5) ? ERR : 0; }
        *log = 6;
        return build_seq_table(t, 2, ML_def, 52, 6) ? ERR : 0;
    }
    if (mode == 1) {
        if (n < 1 || src[0] > max_sym_k[kind]) return ERR;
        *log = 0;
        t[0].nbits = 0; t[0].next = 0;
        sym_info(kind, src[0], &t[0].base, &t[0].extra);
        return 1;
    }
    if (mode == 2) {
        int ms = max_sym_k[kind], lg;
        long used = read_ncount(norm, &ms, &lg, src, n);
        if (used < 0 || lg > max_log_k[kind]) return ERR;
        if (build_seq_table(t, kind, norm, ms, lg)) return ERR;
        *log = lg;
        return used;
    }
    return have_prev ? 0 : ERR; /* repeat */
}

static long decode_block(zctx *z, const uint8_t *src, size_t n, uint8_t *dst, size_t op, size_t cap)
{
    size_t lit_size = 0, lit_pos = 0, out = op;
    long used = decode_literals(z, src, n, &lit_size);
    const uint8_t *ip;
    size_t left;
    uint32_t nseq;
    if (used < 0) return ERR;
    ip = src + used;
    left = n - (size_t)used;
    if (left < 1) return ERR;
    nseq = *ip++; left--;
    if (nseq == 0) { if (left != 0) return ERR; }
    else {
        brd b;
        uint32_t sl, so, sm, i;
        int modes;
        if (nseq > 0x7F) {
            if (nseq == 0xFF) { if (left < 2) return ERR; nseq = (uint32_t)ip[0] + ((uint32_t)ip[1] << 8) + 0x7F00; ip += 2; left -= 2; }
            else { if (left < 1) return ERR; nseq = ((nseq - 0x80) << 8) + *ip++; left--; }
        }
        if (left < 1) return ERR;
        modes = *ip++; left--;
        {
            long u = read_seq_table(z->ll, &z->ll_log, 0, modes >> 6, ip, left, z->fse_valid);
            if (u < 0) return ERR;
            ip += u; left -= (size_t)u;
            u = read_seq_table(z->of, &z->of_log, 1, (modes >> 4) & 3, ip, left, z->fse_valid);
            if (u < 0) return ERR;
            ip += u; left -= (size_t)u;
            u = read_seq_table(z->ml, &z->ml_log, 2, (modes >> 2) & 3, ip, left, z->fse_valid);
            if (u < 0) return ERR;
            ip += u; left -= (size_t)u;
        }
        z->fse_valid = 1;
        if (br_init(&b, ip, left)) return ERR;
        sl = br_read(&b, z->ll_log);
        so = br_read(&b, z->of_log);
        sm = br_read(&b, z->ml_log);
        for (i = 0; i < nseq; i++) {
            const seq_ent *el = &z->ll[sl], *eo = &z->of[so], *em = &z->ml[sm];
            uint32_t offset, mlen, llen;
            const int ll0 = (el->base == 0);
            if (eo->extra > 1) {
                offset = eo->base + br_read(&b, eo->extra);
                z->rep[2] = z->rep[1]; z->rep[1] = z->rep[0]; z->rep[0] = offset;
            } else if (eo->extra == 0) {
                if (!ll0) offset = z->rep[0];
                else { offset = z->rep[1]; z->rep[1] = z->rep[0]; z->rep[0] = offset; }
            } else {
                uint32_t idx = eo->base + (uint32_t)ll0 + br_read(&b, 1);
                uint32_t tmp = (idx == 3) ? z->rep[0] - 1 : z->rep[idx];
                tmp += !tmp; /* 0 is not valid: corrupted input is forced to offset 1 */
                if (idx != 1) z->rep[2] = z->rep[1];
                z->rep[1] = z->rep[0];
                z->rep[0] = offset = tmp;
            }
            mlen = em->base + (em->extra ? br_read(&b, em->extra) : 0);
            llen = el->base + (el->extra ? br_read(&b, el->extra) : 0);
            if (i + 1 < nseq) { /* state updates: LL, ML, OF */
                sl = el->next + br_read(&b, el->nbits);
                sm = em->next + br_read(&b, em->nbits);
                so = eo->next + br_read(&b, eo->nbits);
            }
            if (getenv("CRYO_ORACLE_TRACE")) fprintf(stderr, "seq out=%zu ll=%u ml=%u off=%u osym_extra=%u\n", out, llen, mlen, offset, (unsigned)eo->extra);
            /* execute */
            if (llen > lit_size - lit_pos) return ERR;
            if ((size_t)llen + mlen > cap - out) return ERR;
            memcpy(dst + out, z->lit + lit_pos, llen);
            out += llen; lit_pos += llen;
            if (offset > out) return ERR;
            { uint32_t k; for (k = 0; k < mlen; k++) dst[out + k] = dst[out + k - offset]; }
            out += mlen;
        }
        if (b.over || b.pos != 0) return ERR; /* the bitstream must be consumed exactly */
    }
    if (lit_size - lit_pos > cap - out) return ERR;
    memcpy(dst + out, z->lit + lit_pos, lit_size - lit_pos);
    out += lit_size - lit_pos;
    return (long)(out - op);
}

/* ------------------------------------------------------------------ XXH64 (content checksum) */
static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t xxh64(const uint8_t *p, size_t len)
{
    const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull,
                   P4 = 9650029242287828579ull, P5 = 2870177450012600261ull;
    const uint8_t *end = p + len;
    uint64_t h;
#define RND(a, v) (rotl64((a) + (v) * P2, 31) * P1)
#define MRG(h, v) (((h) ^ RND(0, v)) * P1 + P4)
    if (len >= 32) {
        uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
        do { v1 = RND(v1, rd64(p)); v2 = RND(v2, rd64(p + 8)); v3 = RND(v3, rd64(p + 16)); v4 = RND(v4, rd64(p + 24)); p += 32; }
        while (p + 32 <= end);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = MRG(h, v1); h = MRG(h, v2); h = MRG(h, v3); h = MRG(h, v4);
    } else h = P5;
    h += len;
    while (p + 8 <= end) { h ^= RND(0, rd64(p)); h = rotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (*p++) * P5; h = rotl64(h, 11) * P1; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
#undef RND
#undef MRG
    return h;
}

/* ------------------------------------------------------------------ frames */
size_t cryo_oracle_zstd_bound(size_t n)
{
    return n + (n >> 8) + (n < (128u << 10) ? ((128u << 10) - n) >> 11 : 0);
}

static zctx g_ctx; /* the oracle is single-threaded test code */

long cryo_oracle_zstd_decompress(const uint8_t *src, size_t csize, uint8_t *dst, size_t cap)
{
    size_t ip = 0, op = 0;
    zctx *z = &g_ctx;
    while (csize - ip >= 5) { /* ZSTD_startingInputLength */
        uint32_t magic;
        if (csize - ip >= 8 && ((rd32(src + ip) & 0xFFFFFFF0u) == 0x184D2A50u)) { /* skippable frame */
            uint32_t sz = rd32(src + ip + 4);
            if ((uint64_t)sz + 8 > csize - ip) return ERR;
            ip += 8 + sz;
            continue;
        }
        magic = rd32(src + ip);
        if (magic != 0xFD2FB528u) return ERR;
        {
            const uint8_t fhd = src[ip + 4];
            const int single = (fhd >> 5) & 1, did = fhd & 3, fcs_flag = fhd >> 6, has_ck = (fhd >> 2) & 1;
            static const int did_sz[4] = {0, 1, 2, 4}, fcs_sz[4] = {0, 2, 4, 8};
            size_t hsz = 5 + !single + did_sz[did] + (fcs_flag ? fcs_sz[fcs_flag] : single);
            uint64_t fcs = ~0ull;
            size_t frame_start = op, p;
            if (fhd & 0x08) return ERR; /* reserved bit */
            if (csize - ip < hsz) return ERR;
            p = ip + 5;
            if (!single) { if ((src[p] >> 3) + 10 > 31) return ERR; p++; }
            if (did) {
                uint32_t id = 0; int k;
                for (k = 0; k < did_sz[did]; k++) id |= (uint32_t)src[p + k] << (8 * k);
                if (id != 0) return ERR; /* a dictionary would be needed */
                p += did_sz[did];
            }
            if (fcs_flag == 0) { if (single) fcs = src[p]; }
            else if (fcs_flag == 1) fcs = (uint64_t)(src[p] | (src[p + 1] << 8)) + 256;
            else if (fcs_flag == 2) fcs = rd32(src + p);
            else fcs = rd64(src + p);
            ip += hsz;
            z->huf_valid = 0; z->fse_valid = 0;
            z->rep[0] = 1; z->rep[1] = 4; z->rep[2] = 8;
            for (;;) {
                uint32_t bh, bsize;
                int last, type;
                if (csize - ip < 3) return ERR;
                bh = src[ip] | (src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16);
                ip += 3;
                last = bh & 1; type = (bh >> 1) & 3; bsize = bh >> 3;
                if (type == 3) return ERR;
                if (type == 1) {
                    if (csize - ip < 1) return ERR;
                    if (bsize > cap - op) return ERR;
                    memset(dst + op, src[ip], bsize);
                    ip += 1; op += bsize;
                } else {
                    if (bsize > csize - ip) return ERR;
                    if (type == 0) {
                        if (bsize > cap - op) return ERR;
                        memcpy(dst + op, src + ip, bsize);
                        op += bsize;
                    } else {
                        long r;
                        if (bsize >= ZBLOCK_MAX) return ERR;
                        r = decode_block(z, src + ip, bsize, dst, op, cap);
                        if (r < 0) return ERR;
                        op += (size_t)r;
                    }
                    ip += bsize;
                }
                if (last) break;
            }
            if (fcs != ~0ull && (uint64_t)(op - frame_start) != fcs) return ERR;
            if (has_ck) {
                if (csize - ip < 4) return ERR;
                if ((uint32_t)xxh64(dst + frame_start, op - frame_start) != rd32(src + ip)) return ERR;
                ip += 4;
            }
        }
    }
    if (ip != csize) return ERR; /* input not entirely consumed */
    return (long)op;
}
