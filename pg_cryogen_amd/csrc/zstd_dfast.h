/*
 * zstd_dfast.h -- the `dfast` strategy's match finder (libzstd 1.4.8 ZSTD_compressBlock_doubleFast, no
 * dictionary; zstd levels 3 and 4 at cryo block sizes), 64 search positions per step.  Included by
 * zstd_enc.hip inside its namespace; the entropy stage behind it is the one the `fast` levels use.
 *
 * Replaces the match-finding half of ZSTD_compress(dst, bound, src, B, level) (reference
 * compression.c:102-104) for level 3 / 4; restated for the CPU in oracle/zstd_enc_oracle.c (block_dfast).
 *
 * The algorithm: a long table hashed on 8 bytes and a short one hashed on minMatch bytes; every visited
 * position reads its slot of both, then writes its own index to both.  Tests in order: repeat offset at
 * ip+1, long candidate at ip (8 bytes equal), short candidate at ip (4 bytes equal; then the long table is
 * also tried at ip+1 and wins if it matches).  No hit: ip += ((ip - anchor) >> 8) + 1.
 *
 * Both tables are too large for LDS (2^16 + 2^15 ... 2^18 + 2^18 entries), so they live in the workgroup's
 * global workspace and a position costs dependent trips to L2/HBM: input -> table -> candidate.  Taking the
 * next 64 positions of the walk at once (one per lane; they are known in advance as long as nothing is
 * found: same step while (ip - anchor) >> 8 is unchanged) pays those trips once per 64 positions.  A lane
 * must see what earlier lanes of the same step wrote to its slots; instead of resolving that, the step is
 * cut short in front of the first lane that could share a slot with an earlier one (a small LDS array
 * indexed by the slot's low bits, marked with lane numbers, finds those -- conservatively), so every lane
 * that stays reads exactly the table state the serial walk would see.  The first lane that finds anything
 * ends the step; only lanes up to it write their index to the tables.
 */
#pragma once

constexpr uint32_t kDfMark = 8192; /* slots of the duplicate filter (bytes of LDS) */

__device__ inline uint64_t ld64v(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ inline uint32_t ld32v(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

__device__ inline uint32_t hash8_v(uint64_t v, int hlog) { return (uint32_t)((v * 0xCF1BBCDCB7A56463ull) >> (64 - hlog)); }
__device__ inline uint32_t hashs_v(uint64_t v, int hlog, int mls)
{
    switch (mls) {
    default:
    case 4: return ((uint32_t)v * 2654435761u) >> (32 - hlog);
    case 5: return (uint32_t)(((v << 24) * 889523592379ull) >> (64 - hlog));
    case 6: return (uint32_t)(((v << 16) * 227718039650203ull) >> (64 - hlog));
    case 7: return (uint32_t)(((v << 8) * 58295818150454627ull) >> (64 - hlog));
    }
}

__device__ inline uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)v, o, 64); v = x < v ? x : v; }
    return uni(v);
}

/* how many bytes before a and b are equal, at most lim (the library's catch-up loop, 64 bytes per step) */
__device__ inline uint32_t count_back(const uint8_t *a, const uint8_t *b, uint32_t lim, uint32_t lane)
{
    uint32_t done = 0;
    for (;;) {
        const uint32_t k = done + lane;
        const bool eq = k < lim && a[-1 - (int)k] == b[-1 - (int)k];
        const unsigned long long neq = __ballot(!eq);
        if (neq != 0ull) return done + ctz64(neq);
        done += 64u;
    }
}

/* marks `slot` with the lane number; returns the lowest lane that shares a marked slot with another lane (64: none) */
__device__ inline uint32_t first_shared_slot(uint8_t *mark, uint32_t slot, bool on, uint32_t lane)
{
    if (on) mark[slot] = (uint8_t)lane;
    asm volatile("" ::: "memory"); /* the read below must come from LDS, not from this lane's own store */
    __builtin_amdgcn_wave_barrier();
    const uint32_t r = on ? (uint32_t)mark[slot] : lane;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    return wave_min_u32(r != lane ? (r < lane ? r : lane) : 64u);
}

/* ZSTD_compressBlock_doubleFast.  Indexes are the library's (base = src - 1: the first input byte is 1). */
__device__ uint32_t block_dfast_batch(uint32_t *tl, uint32_t *ts, uint8_t *mark, const CPar &cp, const uint8_t *base,
                                      const uint8_t *istart, uint32_t n, uint32_t *rep, uint8_t *ws, SeqStore &ss,
                                      uint32_t dict_limit, uint32_t lane)
{
    const int hl = cp.hlog, hs = cp.clog, mls = cp.mml < 4 ? 4 : (cp.mml > 7 ? 7 : cp.mml);
    uint32_t ip = (uint32_t)(istart - base), anchor = ip;
    const uint32_t iend = ip + n, ilimit = iend - 8u;
    const uint32_t max_dist = 1u << cp.wlog;
    const uint32_t prefix_idx = (iend - dict_limit > max_dist) ? iend - max_dist : dict_limit;
    uint32_t off1 = rep[0], off2 = rep[1], saved = 0;
    if (ip == prefix_idx) ip++;
    {
        const uint32_t wlow = (ip - dict_limit > max_dist) ? ip - max_dist : dict_limit;
        const uint32_t max_rep = ip - wlow;
        if (off2 > max_rep) { saved = off2; off2 = 0; }
        if (off1 > max_rep) { saved = off1; off1 = 0; }
    }
    while (ip < ilimit) {
        const uint32_t st = ((ip - anchor) >> 8) + 1u;
        const uint32_t p = ip + lane * st;
        bool valid = p < ilimit && ((p - anchor) >> 8) + 1u == st;
        const uint64_t v8 = valid ? ld64v(base + p) : 0ull;
        const uint32_t h2 = hash8_v(v8, hl), h = hashs_v(v8, hs, mls);
        {
            const uint32_t f1 = first_shared_slot(mark, h2 & (kDfMark - 1u), valid, lane);
            const uint32_t f2 = first_shared_slot(mark, h & (kDfMark - 1u), valid, lane);
            const uint32_t f = f1 < f2 ? f1 : f2;
            valid = valid && lane <= f;
        }
        const uint32_t cl = valid ? tl[h2] : 0u, cs = valid ? ts[h] : 0u;
        const bool rephit = valid && off1 > 0u && ld32v(base + p + 1u - off1) == (uint32_t)(v8 >> 8);
        const bool longhit = valid && cl > prefix_idx && ld64v(base + cl) == v8;
        const bool shorthit = valid && cs > prefix_idx && ld32v(base + cs) == (uint32_t)v8;
        const unsigned long long repm = __ballot(rephit), longm = __ballot(longhit);
        const unsigned long long hitm = repm | longm | __ballot(shorthit);
        const uint32_t T = hitm ? ctz64(hitm) : 63u;
        const uint32_t ncommit = hitm ? T + 1u : (uint32_t)__builtin_popcountll(__ballot(valid));
        if (lane < ncommit) { tl[h2] = p; ts[h] = p; }
        if (!hitm) { ip += ncommit * st; continue; }

        const uint32_t cur = __builtin_amdgcn_readlane(p, T);
        uint32_t mlen;
        if ((repm >> T) & 1ull) {
            ip = cur + 1u;
            mlen = count_match(base + ip + 4u, base + ip + 4u - off1, base + iend, lane) + 4u;
            store_seq(ws, ss, ip - anchor, base + anchor, 0, mlen - 3u, lane);
        } else {
            uint32_t m;
            if ((longm >> T) & 1ull) {
                m = __builtin_amdgcn_readlane(cl, T);
                ip = cur;
                mlen = count_match(base + ip + 8u, base + m + 8u, base + iend, lane) + 8u;
            } else {
                const uint64_t v9 = ld64u(base + cur + 1u);
                const uint32_t hl3 = hash8_v(v9, hl);
                const uint32_t mil3 = uni(tl[hl3]); /* after this step's own writes, as in the library */
                if (lane == 0) tl[hl3] = cur + 1u;
                if (mil3 > prefix_idx && ld64u(base + mil3) == v9) {
                    m = mil3;
                    ip = cur + 1u;
                    mlen = count_match(base + ip + 8u, base + m + 8u, base + iend, lane) + 8u;
                } else {
                    m = __builtin_amdgcn_readlane(cs, T);
                    ip = cur;
                    mlen = count_match(base + ip + 4u, base + m + 4u, base + iend, lane) + 4u;
                }
            }
            const uint32_t offset = ip - m;
            const uint32_t la = ip - anchor, lm = m - prefix_idx;
            const uint32_t back = count_back(base + ip, base + m, la < lm ? la : lm, lane);
            ip -= back;
            mlen += back;
            off2 = off1;
            off1 = offset;
            store_seq(ws, ss, ip - anchor, base + anchor, offset + 2u, mlen - 3u, lane);
        }
        ip += mlen;
        anchor = ip;
        if (ip <= ilimit) {
            const uint32_t ins = cur + 2u;
            const uint64_t va = ld64u(base + ins), vb = ld64u(base + ip - 2u), vc = ld64u(base + ip - 1u);
            if (lane == 0) {
                tl[hash8_v(va, hl)] = ins;
                tl[hash8_v(vb, hl)] = ip - 2u;
                ts[hashs_v(va, hs, mls)] = ins;
                ts[hashs_v(vc, hs, mls)] = ip - 1u;
            }
            while (ip <= ilimit && off2 > 0u && ld32u(base + ip) == ld32u(base + ip - off2)) {
                const uint32_t rlen = count_match(base + ip + 4u, base + ip + 4u - off2, base + iend, lane) + 4u;
                const uint32_t t = off2; off2 = off1; off1 = t;
                if (lane == 0) {
                    const uint64_t v = ld64v(base + ip);
                    ts[hashs_v(v, hs, mls)] = ip;
                    tl[hash8_v(v, hl)] = ip;
                }
                store_seq(ws, ss, 0, base + anchor, 0, rlen - 3u, lane);
                ip += rlen;
                anchor = ip;
            }
        }
    }
    rep[0] = off1 ? off1 : saved;
    rep[1] = off2 ? off2 : saved;
    return iend - anchor;
}
