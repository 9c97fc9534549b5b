/*
 * lz4_oracle.c -- CPU ORACLE (test infrastructure, see cryo_oracle.h).
 *
 * Restates, in index arithmetic, the LZ4 block codec as liblz4 1.9.3 runs it
 * for the reference's two call shapes:
 *   compression.c:67-72  LZ4_compress_fast(data, out, CRYO_BLCKSZ, LZ4_compressBound(CRYO_BLCKSZ), accel)
 *   compression.c:84     LZ4_decompress_safe(compressed, out, compressed_size, CRYO_BLCKSZ)
 * liblz4 is a third-party dependency of the reference (Makefile:5 -llz4), not
 * vendored and not version-pinned there; parity is pinned to liblz4 1.9.3 by
 * tests/golden (see tests/golden/make_golden.py).
 *
 * Encoder facts that fix the bytes (checked against the library by the tests):
 *   - dstCapacity >= bound  -> the "not limited" output path (no output checks)
 *   - n <  65547            -> 16-bit position table, 8192 entries, 4-byte hash
 *   - n >= 65547            -> 32-bit position table, 4096 entries, 5-byte hash
 *   - accel <= 0 -> 1, accel > 65537 -> 65537 (lz4.h:189-190)
 *   - table starts zeroed, positions are offsets from the start of the input
 */
#include "cryo_oracle.h"
#include <string.h>

enum {
    MINMATCH = 4,
    MFLIMIT = 12,
    LASTLITERALS = 5,
    MINLENGTH = MFLIMIT + 1,
    MAXDIST = 65535,
    SKIPTRIGGER = 6,
    LIMIT64K = 65536 + (MFLIMIT - 1),
    MAXINPUT = 0x7E000000
};

static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

size_t cryo_oracle_lz4_bound(size_t n)
{
    return n > (size_t)MAXINPUT ? 0 : n + n / 255 + 16;
}

/* hash of the bytes at position p: 13-bit/4-byte for the 16-bit table, 12-bit/5-byte otherwise */
static uint32_t hash_at(const uint8_t *src, size_t p, int small)
{
    if (small) return (rd32(src + p) * 2654435761u) >> (32 - 13);
    return (uint32_t)(((rd64(src + p) << 24) * 889523592379ull) >> (64 - 12));
}

static size_t put_len(uint8_t *dst, size_t o, size_t len)
{
    for (; len >= 255; len -= 255) dst[o++] = 255;
    dst[o++] = (uint8_t)len;
    return o;
}

size_t cryo_oracle_lz4_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, int accel)
{
    uint32_t table[8192];
    const int small = n < (size_t)LIMIT64K;
    size_t ip = 0, anchor = 0, op = 0;
    size_t mflimit_p1, matchlimit;
    uint32_t fwd_h;

    if (n > (size_t)MAXINPUT || cap < cryo_oracle_lz4_bound(n)) return 0;
    if (accel < 1) accel = 1;
    if (accel > 65537) accel = 65537;
    memset(table, 0, sizeof table);

    if (n < (size_t)MINLENGTH) goto tail;
    mflimit_p1 = n - MFLIMIT + 1;
    matchlimit = n - LASTLITERALS;

    table[hash_at(src, 0, small)] = 0;
    ip = 1;
    fwd_h = hash_at(src, ip, small);

    for (;;) {
        size_t match, tok;
        /* search: probe with a step that grows after (64/accel) misses */
        {
            size_t fwd = ip;
            uint32_t step = 1, nb = (uint32_t)accel << SKIPTRIGGER;
            for (;;) {
                uint32_t h = fwd_h;
                size_t cur = fwd;
                ip = fwd;
                fwd += step;
                step = nb++ >> SKIPTRIGGER;
                if (fwd > mflimit_p1) goto tail;
                match = table[h];
                fwd_h = hash_at(src, fwd, small);
                table[h] = (uint32_t)cur;
                if (!small && match + MAXDIST < cur) continue;
                if (rd32(src + match) == rd32(src + ip)) break;
            }
        }
        /* extend backwards */
        while (ip > anchor && match > 0 && src[ip - 1] == src[match - 1]) { ip--; match--; }

        /* literals */
        {
            size_t lit = ip - anchor;
            tok = op++;
            if (lit >= 15) { dst[tok] = 15 << 4; op = put_len(dst, op, lit - 15); }
            else dst[tok] = (uint8_t)(lit << 4);
            memcpy(dst + op, src + anchor, lit);
            op += lit;
        }
        for (;;) {
            /* offset + match length */
            size_t a = ip + MINMATCH, b = match + MINMATCH, ml;
            dst[op++] = (uint8_t)(ip - match);
            dst[op++] = (uint8_t)((ip - match) >> 8);
            while (a < matchlimit && src[a] == src[b]) { a++; b++; }
            ml = a - (ip + MINMATCH);
            ip = a;
            if (ml >= 15) { dst[tok] += 15; op = put_len(dst, op, ml - 15); }
            else dst[tok] += (uint8_t)ml;
            anchor = ip;
            if (ip >= mflimit_p1) goto tail;

            table[hash_at(src, ip - 2, small)] = (uint32_t)(ip - 2);
            /* immediate re-test at ip */
            {
                uint32_t h = hash_at(src, ip, small);
                match = table[h];
                table[h] = (uint32_t)ip;
                if ((small || match + MAXDIST >= ip) && rd32(src + match) == rd32(src + ip)) {
                    tok = op++;
                    dst[tok] = 0;
                    continue;
                }
            }
            break;
        }
        fwd_h = hash_at(src, ++ip, small);
    }

tail:
    {
        size_t lit = n - anchor;
        if (lit >= 15) { dst[op++] = 15 << 4; op = put_len(dst, op, lit - 15); }
        else dst[op++] = (uint8_t)(lit << 4);
        memcpy(dst + op, src + anchor, lit);
        op += lit;
    }
    return op;
}

/*
 * LZ4_decompress_safe semantics (full-block, no dictionary): accept/reject
 * decisions follow liblz4 1.9.3's bounds rules -
 *   - a literal run that reaches within 12 bytes of the output end or within
 *     8 bytes of the input end must be the final one and end the input exactly;
 *   - literal-length extension bytes may not start at/after iend-15;
 *   - match-length extension bytes may not run to iend-4 or beyond;
 *   - offset must not reach before the output start;
 *   - a match must end at least 5 bytes before the output capacity;
 *   - offset 0 is not rejected by 1.9.3: it reproduces as zero bytes.
 */
long cryo_oracle_lz4_decompress(const uint8_t *src, size_t csize, uint8_t *dst, size_t cap)
{
    size_t ip = 0, op = 0;
    if (cap == 0) return (csize == 1 && src[0] == 0) ? 0 : -1;
    if (csize == 0) return -1;
    for (;;) {
        uint32_t token = src[ip++];
        size_t len = token >> 4, off, i;
        if (len == 15) {
            uint32_t s;
            if (ip + 15 >= csize) return -1;
            do {
                s = src[ip++];
                len += s;
                if (ip + 15 >= csize) break;    /* library keeps the partial length; rejected below */
            } while (s == 255);
        }
        if (op + len + MFLIMIT > cap || ip + len + 8 > csize) {
            if (ip + len != csize || op + len > cap) return -1;
            memmove(dst + op, src + ip, len);
            op += len;
            return (long)op;
        }
        memcpy(dst + op, src + ip, len);
        ip += len; op += len;

        off = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
        ip += 2;
        len = token & 15;
        if (len == 15) {
            uint32_t s;
            do {
                s = src[ip++];
                len += s;
                if (ip + 4 >= csize) return -1;
            } while (s == 255);
        }
        len += MINMATCH;
        if (off > op) return -1;
        if (op + len + LASTLITERALS > cap) return -1;
        if (off == 0) { memset(dst + op, 0, len); op += len; continue; }
        if (off >= len) memcpy(dst + op, dst + op - off, len);
        else for (i = 0; i < len; i++) dst[op + i] = dst[op + i - off];
        op += len;
    }
}
