/*
 * zstd_dfast.h -- the match finders of zstd's `dfast` and `fast` strategies (libzstd 1.4.8
 * ZSTD_compressBlock_doubleFast / ZSTD_compressBlock_fast, no dictionary; levels 3, 4 and -5..2 at cryo
 * block sizes) over tables in global memory, many search positions per step.  Included by zstd_enc.hip
 * inside its namespace, in front of the entropy stage.
 *
 * Replaces the match-finding half of ZSTD_compress(dst, bound, src, B, level) (reference
 * compression.c:102-104); restated for the CPU in oracle/zstd_enc_oracle.c (block_dfast, block_fast).
 *
 * dfast: a long table hashed on 8 bytes and a short one hashed on minMatch bytes; every visited
 * position reads its slot of both, then writes its own index to both.  Tests in order: repeat offset at
 * ip+1, long candidate at ip (8 bytes equal), short candidate at ip (4 bytes equal; then the long table is
 * also tried at ip+1 and wins if it matches).  No hit: ip += ((ip - anchor) >> 8) + 1.
 *
 * Both tables are too large for LDS (2^16 + 2^15 ... 2^18 + 2^18 entries), so they live in the workgroup's
 * global workspace and a position costs dependent trips to memory: input -> table slots -> candidates.  The
 * next positions of the walk are known in advance as long as nothing is found (same step while
 * (ip - anchor) >> 8 is unchanged), so a step takes up to 64 of them at once, one per lane (32 by default:
 * measured best), and pays those trips once.  A lane must see what earlier lanes of the same step wrote to
 * its slots; instead of resolving that, the step is cut short in front of the first lane that could share a
 * slot with an earlier one (a small LDS array indexed by the slot's low bits, marked with lane numbers, finds
 * those -- conservatively), so every lane that stays reads exactly the table state the serial walk would
 * see.  The first lane that finds anything ends the step; only lanes up to it write their index to the
 * tables.  Per sequence the trips are: table slots -> the three candidates at once -> match extension both
 * ways at once -> one tail trip (complementary insertions, immediate-repeat check, the next step's input and
 * the next literal run together).  The long-table lookup at ip+1 that follows a short hit is the next lane's
 * own lookup when the step is 1, so it costs no trip.
 * Measured (65536 x 128 KiB text-like rows, one MI355X, 4096 waves in flight): level 1 19.5, level 2 12.4,
 * level 3 10.3, level 4 5.1 GB/s (three times the sequences); thousands of cycles per sequence, shared between
 * memory latency and the rate of scattered accesses the waves put on HBM (tables: 0.06-2 MiB per wave, far
 * beyond L2/MALL).  profiles/r01_zstd_encoders.txt has the history and the phase split.
 */
#pragma once

constexpr uint32_t kDfMark = 4096; /* slots of the duplicate filter (bytes of LDS, over the entropy stage's scratch) */

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

/* marks `slot` with the lane number; returns how many leading lanes have slots no earlier lane shares
 * (conservative: slots are the table slots' low bits) */
__device__ inline uint32_t distinct_prefix(uint8_t *mark, uint32_t slot, bool on, uint32_t lane)
{
    if (on) mark[slot] = (uint8_t)lane;
    asm volatile("" ::: "memory"); /* the read below must come from LDS, not from this lane's own store */
    __builtin_amdgcn_wave_barrier();
    const uint32_t r = on ? (uint32_t)mark[slot] : lane;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long losers = __ballot(r != lane);
    if (!losers) return 64u;
    /* lanes in front of the first loser have distinct slots; the first loser itself is fine when the lane that
     * won its slot comes later */
    const uint32_t l = ctz64(losers);
    return (uint32_t)__builtin_amdgcn_readlane(r, l) > l ? l + 1u : l;
}

/* literal run of a sequence into the block's literal buffer: the first 64 bytes were loaded ahead (litv) */
__device__ inline void store_seq_pre(uint8_t *ws, SeqStore &ss, uint32_t ll, uint32_t litv, const uint8_t *lit,
                                     uint32_t offcode, uint32_t mlbase, uint32_t lane)
{
    uint8_t *lits = ws + kWsLit;
    if (lane < ll) lits[ss.nlit + lane] = (uint8_t)litv;
    for (uint32_t i = 64u + lane; i < ll; i += 64u) lits[ss.nlit + i] = lit[i];
    ss.nlit += ll;
    if (ll > 0xFFFFu) { ss.long_kind = 1; ss.long_pos = ss.nseq; }
    if (mlbase > 0xFFFFu) { ss.long_kind = 2; ss.long_pos = ss.nseq; }
    if (lane == 0) reinterpret_cast<uint2 *>(ws + kWsSeq)[ss.nseq] = make_uint2(offcode + 1u, (ll & 0xFFFFu) | (mlbase << 16));
    ss.nseq++;
}

/* ZSTD_compressBlock_doubleFast.  Indexes are the library's (base = src - 1: the first input byte is 1).
 * Trips to memory per sequence: [input of the step, loaded with the previous sequence's tail] -> table slots
 * -> candidates (repeat offset, long, short: one trip, branch-free) -> match extension both ways -> tail
 * (complementary insertions, immediate repeat check, next step's input, next literal run). */
template <bool PROF>
__device__ uint32_t block_dfast_batch(uint32_t *tl, uint32_t *ts, uint8_t *mark, const CPar &cp, const uint8_t *base,
                                      const uint8_t *istart, uint32_t n, uint32_t *rep, uint8_t *ws, SeqStore &ss,
                                      uint32_t dict_limit, uint32_t lane, uint32_t W, unsigned long long *prof)
{
    /* CRYO_ZSTD_STATS (PROF): cycles per phase and event counts; compiled out of the production kernel */
    unsigned long long pt[PROF ? 8 : 1] = {0}, pn[PROF ? 6 : 1] = {0}, t0 = PROF ? __builtin_amdgcn_s_memtime() : 0;
#define DFT(k) do { if constexpr (PROF) { const unsigned long long t = __builtin_amdgcn_s_memtime(); pt[k] += t - t0; t0 = t; } } while (0)
#define DFN(k, v) do { if constexpr (PROF) pn[k] += (v); } while (0)
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
    uint32_t litv = anchor + lane < iend ? base[anchor + lane] : 0u; /* literal bytes anchor .. anchor+63 */
    uint64_t v8n = 0;    /* input of the next step, when loaded ahead */
    bool have = false;
    while (ip < ilimit) {
        const uint32_t st = ((ip - anchor) >> 8) + 1u;
        const uint32_t p = ip + lane * st;
        bool valid = lane < W && p < ilimit && ((p - anchor) >> 8) + 1u == st;
        const uint64_t v8 = have ? v8n : (valid ? ld64v(base + p) : 0ull);
        have = false;
        const uint32_t h2 = hash8_v(v8, hl), h = hashs_v(v8, hs, mls);
        if constexpr (PROF) asm volatile("" :: "v"(h2), "v"(h));
        DFT(0);
        {
            const uint32_t n1 = distinct_prefix(mark, h2 & (kDfMark - 1u), valid, lane);
            const uint32_t n2 = distinct_prefix(mark, h & (kDfMark - 1u), valid, lane);
            valid = valid && lane < (n1 < n2 ? n1 : n2);
        }
        DFT(1);
        const uint32_t cl = valid ? tl[h2] : 0u, cs = valid ? ts[h] : 0u;
        if constexpr (PROF) asm volatile("" :: "v"(cl), "v"(cs));
        DFT(2);
        /* the three candidates in one trip: addresses of lanes that have no candidate point at ip */
        const bool rc = valid && off1 > 0u, lc = valid && cl > prefix_idx, sc = valid && cs > prefix_idx;
        const uint32_t rv = ld32v(base + (rc ? p + 1u - off1 : ip));
        const uint64_t lv = ld64v(base + (lc ? cl : ip));
        const uint32_t sv = ld32v(base + (sc ? cs : ip));
        const bool rephit = rc && rv == (uint32_t)(v8 >> 8);
        const bool longhit = lc && lv == v8;
        const bool shorthit = sc && sv == (uint32_t)v8;
        const unsigned long long validm = __ballot(valid);
        const unsigned long long repm = __ballot(rephit), longm = __ballot(longhit);
        const unsigned long long hitm = repm | longm | __ballot(shorthit);
        const uint32_t T = hitm ? ctz64(hitm) : 63u;
        const uint32_t ncommit = hitm ? T + 1u : (uint32_t)__builtin_popcountll(validm);
        DFT(3);
        if (lane < ncommit) { tl[h2] = p; ts[h] = p; }
        DFN(0, 1); DFN(1, ncommit);
        if (!hitm) { ip += ncommit * st; continue; }
        DFN(2, 1);

        const uint32_t cur = __builtin_amdgcn_readlane(p, T);
        uint32_t mlen, ll, offcode, m, known, blim;
        if ((repm >> T) & 1ull) {
            ip = cur + 1u;
            m = ip - off1;
            known = 4u;
            blim = 0;
            offcode = 0;
        } else {
            if ((longm >> T) & 1ull) {
                m = __builtin_amdgcn_readlane(cl, T);
                ip = cur;
                known = 8u;
            } else {
                /* long table at ip+1.  With step 1 the next lane has read that slot and compared its candidate already
                 * (its slot differs from every slot written by this step); otherwise look it up now */
                uint32_t hl3, mil3;
                bool hit3;
                if (st == 1u && T < 63u && ((validm >> (T + 1u)) & 1ull)) {
                    hl3 = __builtin_amdgcn_readlane(h2, T + 1u);
                    mil3 = __builtin_amdgcn_readlane(cl, T + 1u);
                    hit3 = (longm >> (T + 1u)) & 1ull;
                } else {
                    const uint64_t v9 = ld64u(base + cur + 1u);
                    hl3 = hash8_v(v9, hl);
                    mil3 = uni(tl[hl3]); /* after this step's own writes, as in the library */
                    hit3 = mil3 > prefix_idx && ld64u(base + mil3) == v9;
                }
                if (lane == 0) tl[hl3] = cur + 1u;
                if (hit3) { m = mil3; ip = cur + 1u; known = 8u; }
                else { m = __builtin_amdgcn_readlane(cs, T); ip = cur; known = 4u; }
            }
            const uint32_t la = ip - anchor, lm = m - prefix_idx;
            blim = la < lm ? la : lm;
            off2 = off1;
            off1 = ip - m;
            offcode = off1 + 2u;
        }
        /* match extension, and with it what the sequence's tail needs whatever the length turns out to be (below 64
         * bytes more), as in block_fast_gbatch: lane j reads 8 bytes at wb + j (A) and wb + 64 + j (A2), wb = ip +
         * known - 2, and the immediate-repeat candidate of a match that ends at ip + known + j */
        const uint32_t wb = ip + known - 2u;
        const bool win_a = wb + lane + 8u <= iend, win_a2 = wb + 64u + lane + 8u <= iend;
        const uint64_t A = ld64v(base + (win_a ? wb + lane : ip)), A2 = ld64v(base + (win_a2 ? wb + 64u + lane : ip));
        const uint32_t r1s = ld32v(base + ((off2 > 0u && ip + known + lane + 4u <= iend) ? ip + known + lane - off2 : ip));
        const uint64_t vas = ld64v(base + cur + 2u);
        uint32_t fwd, back;
        count_both(base + ip + known, base + m + known, base + iend, base + ip, base + m, blim, lane, fwd, back);
        mlen = known + fwd + back;
        ip -= back;
        DFT(4);
        ll = ip - anchor;
        const uint32_t seq_anchor = anchor;
        ip += mlen;
        anchor = ip;
        if (fwd < 64u && ip <= ilimit) {
            auto win64 = [&](uint32_t t) { /* 8 bytes at window position t (index wb + t), t <= 127 */
                const int j = (int)(t & 63u);
                const uint64_t a = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(A >> 32), j, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)A, j, 64);
                const uint64_t b = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(A2 >> 32), j, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)A2, j, 64);
                return t < 64u ? a : b;
            };
            const uint32_t te = fwd + 2u; /* the match ends at window position te */
            const uint32_t r0 = (uint32_t)uni64(win64(te));
            const uint32_t r1 = __builtin_amdgcn_readlane(r1s, fwd);
            if (!(off2 > 0u && r0 == r1)) {
                const uint32_t ins = cur + 2u;
                const uint64_t va = uni64(vas), vb = uni64(win64(fwd)), vc = uni64(win64(fwd + 1u));
                store_seq_pre(ws, ss, ll, litv, base + seq_anchor, offcode, mlen - 3u, lane);
                if (lane == 0) {
                    tl[hash8_v(va, hl)] = ins;
                    tl[hash8_v(vb, hl)] = ip - 2u;
                    ts[hashs_v(va, hs, mls)] = ins;
                    ts[hashs_v(vc, hs, mls)] = ip - 1u;
                }
                const bool nv = lane < W && ip + lane < ilimit;
                const uint32_t tn = te + lane;
                const bool inw = tn <= 127u;
                const uint64_t wv = win64(inw ? tn : 0u);
                v8n = wv;
                if (nv && !inw) v8n = ld64v(base + ip + lane);
                if (!nv) v8n = 0ull;
                {
                    const bool ok = inw && wb + tn + 8u <= iend; /* that window lane was a real read */
                    uint32_t byte = (uint32_t)wv & 0xFFu;
                    if (!ok && ip + lane < iend) byte = base[ip + lane];
                    litv = ip + lane < iend ? byte : 0u;
                }
                have = true;
                DFN(3, 1);
                DFT(6);
                continue;
            }
            DFN(4, 1); /* immediate repeat: the tail loop below */
        }
        /* tail: complementary insertions once, then immediate repeats (offset_2) as long as they match; every
         * round loads the literal bytes and the search input at the new anchor with its other loads */
        bool first = true;
        if (ip > ilimit) store_seq_pre(ws, ss, ll, litv, base + seq_anchor, offcode, mlen - 3u, lane);
        while (ip <= ilimit) {
            /* every load of the round is issued before the first one is waited for (a wave-uniform value is read
             * from lane 0 only after all of them are under way) */
            const bool nv = lane < W && ip + lane < ilimit;
            v8n = nv ? ld64v(base + ip + lane) : 0ull;
            const uint32_t litn = ip + lane < iend ? base[ip + lane] : 0u;
            const uint32_t ins = cur + 2u;
            const uint32_t r0v = ld32v(base + ip), r1v = ld32v(base + ip - off2);
            const uint64_t vav = ld64v(base + (first ? ins : ip)), vbv = ld64v(base + ip - 2u), vcv = ld64v(base + ip - 1u);
            const uint32_t r0 = uni(r0v), r1 = uni(r1v);
            if (first) {
                const uint64_t va = uni64(vav), vb = uni64(vbv), vc = uni64(vcv);
                /* the sequence's own stores go behind the loads of this round */
                store_seq_pre(ws, ss, ll, litv, base + seq_anchor, offcode, mlen - 3u, lane);
                if (lane == 0) {
                    tl[hash8_v(va, hl)] = ins;
                    tl[hash8_v(vb, hl)] = ip - 2u;
                    ts[hashs_v(va, hs, mls)] = ins;
                    ts[hashs_v(vc, hs, mls)] = ip - 1u;
                }
                first = false;
            }
            litv = litn;
            if (!(off2 > 0u && r0 == r1)) { have = true; break; }
            const uint32_t rlen = count_match(base + ip + 4u, base + ip + 4u - off2, base + iend, lane) + 4u;
            const uint32_t t = off2; off2 = off1; off1 = t;
            {
                const uint64_t v = ld64u(base + ip);
                if (lane == 0) { ts[hashs_v(v, hs, mls)] = ip; tl[hash8_v(v, hl)] = ip; }
            }
            store_seq_pre(ws, ss, 0, 0, base + anchor, 0, rlen - 3u, lane);
            ip += rlen;
            anchor = ip;
            DFN(5, 1);
        }
        DFT(6);
    }
    if constexpr (PROF) { if (prof && lane == 0) { for (int k = 0; k < 8; k++) atomicAdd(&prof[8 + k], pt[k]); for (int k = 0; k < 6; k++) atomicAdd(&prof[16 + k], pn[k]); } }
#undef DFT
#undef DFN
    rep[0] = off1 ? off1 : saved;
    rep[1] = off2 ? off2 : saved;
    return iend - anchor;
}

/* marks two slots per lane (read-then-written by one iteration of the `fast` walk); returns how many leading
 * lanes have slots no earlier lane shares.  A lane's own two slots may coincide (the walk reads both before it
 * writes either, and the second write wins: the stores below keep that order). */
__device__ inline uint32_t distinct_prefix2(uint8_t *mark, uint32_t slot0, uint32_t slot1, bool on, uint32_t lane)
{
    if (on) mark[slot0] = (uint8_t)lane;
    asm volatile("" ::: "memory");
    if (on) mark[slot1] = (uint8_t)lane;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const uint32_t r0 = on ? (uint32_t)mark[slot0] : lane, r1 = on ? (uint32_t)mark[slot1] : lane;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long losers = __ballot(r0 != lane || r1 != lane);
    if (!losers) return 64u;
    const uint32_t l = ctz64(losers);
    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane(r0, l), w1 = (uint32_t)__builtin_amdgcn_readlane(r1, l);
    return (w0 >= l && w1 >= l) ? l + 1u : l; /* the first loser stays when the lanes that won its slots come later */
}

/* ZSTD_compressBlock_fast over a table in global memory, many iterations of the search loop per step -- the scheme of
 * block_dfast_batch: an iteration looks at ip0 and ip1 = ip0 + 1 (reads both slots, then writes both) and tests the repeat
 * offset at ip0 + 2, then ip0, then ip1; no hit: both advance by ((ip0 - anchor) >> 7) + stepSize.
 *
 * Round 6: TWO trips to memory per sequence instead of three.  A table entry is index | tag << ib: the bits an index of
 * this frame never uses hold a hash of the four bytes at that position (14 bits at 128 KiB, 11 at 1 MiB).  A candidate whose
 * tag differs cannot match and is never read -- on `wide` a candidate is the last position that hashed to the slot, median
 * age 6.9 KiB, and reading the bytes of 32 of them per step was a second trip and 45 % of the lines fetched
 * (profiles/r06_zstd_enc.txt).  The repeat candidates' addresses do not depend on the table, so they ride with the slots
 * (trip 1).  After it the first lane with a repeat hit (exact) or an equal tag (all but certain) is taken as the match and
 * its verification word rides with the extension / tail loads (trip 2); a tag that lied (2^-14 per look-up) is struck out
 * and the step is decided again.  A step that finds nothing is ONE trip. */
__device__ inline bool wb_ok(uint32_t wb, uint32_t j, uint32_t iend) { return wb + 2u * j + 8u <= iend; }
__device__ inline uint32_t fast_tag(uint32_t v4) { return v4 * 2654435761u; } /* its top bits are the tag */
__device__ uint32_t block_fast_gbatch(uint32_t *table, uint8_t *mark, const CPar &cp, const uint8_t *base,
                                      const uint8_t *istart, uint32_t n, uint32_t *rep, uint8_t *ws, SeqStore &ss,
                                      uint32_t dict_limit, uint32_t lane, uint32_t W)
{
    const int hlog = cp.hlog, mls = cp.mml < 4 ? 4 : (cp.mml > 7 ? 7 : cp.mml);
    const uint32_t step_size = (uint32_t)cp.tlen + (cp.tlen ? 0u : 1u) + 1u;
    const uint32_t ib = (uint32_t)cp.ib, im = (1u << ib) - 1u; /* index bits of an entry; the tag lies above them */
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
    uint32_t litv = anchor + lane < iend ? base[anchor + lane] : 0u;
    uint64_t v0n = 0, v1n = 0;
    bool have = false;
    /* iterations per step: W behind a match; twice as many after every step that found nothing, up to the wave's 64 (a step
     * is a trip to memory whatever its width; rows of hex digits or random bytes are mostly steps without a hit) */
    uint32_t Wc = W;
    while (ip + 1u < ilimit) { /* ip is the walk's ip0; ip1 = ip0 + 1 */
        const uint32_t st = ((ip - anchor) >> 7) + step_size;
        uint32_t i0 = ip + lane * st, inext = i0 + st, nk = Wc;
        if ((((ip + (Wc - 1u) * st) - anchor) >> 7) + step_size != st) {
            /* the walk's stride grows with the literal run (every 128 bytes of it): the lanes' positions are the walk's own
             * recurrence, lane by lane -- a few hundred scalar instructions against a trip to memory per lane group (an
             * incompressible block was one iteration per step from stride 64 on) */
            uint32_t pos = ip;
            nk = 0;
            for (uint32_t k = 0; k < Wc && pos + 1u < ilimit; k++) {
                const uint32_t sk = ((pos - anchor) >> 7) + step_size;
                if (lane == k) { i0 = pos; inext = pos + sk; }
                pos += sk;
                nk = k + 1u;
            }
        }
        bool valid = lane < nk && i0 + 1u < ilimit;
        const uint64_t v0 = have ? v0n : (valid ? ld64v(base + i0) : 0ull);
        const uint64_t v1 = have ? v1n : (valid ? ld64v(base + i0 + 1u) : 0ull);
        have = false;
        const uint32_t h0 = hashs_v(v0, hlog, mls), h1 = hashs_v(v1, hlog, mls);
        valid = valid && lane < distinct_prefix2(mark, h0 & (kDfMark - 1u), h1 & (kDfMark - 1u), valid, lane);
        /* trip 1: the slots and the repeat candidates (rv: the byte in front of the repeat candidate, then its 4 bytes) */
        const bool rc = valid && off1 > 0u;
        const uint32_t e0 = valid ? table[h0] : 0u, e1 = valid ? table[h1] : 0u;
        const uint64_t rv = ld64v(base + (rc ? i0 + 1u - off1 : ip));
        const uint32_t n0 = (i0 & im) | ((fast_tag((uint32_t)v0) >> ib) << ib), n1 = ((i0 + 1u) & im) | ((fast_tag((uint32_t)v1) >> ib) << ib);
        const uint32_t mi0 = e0 & im, mi1 = e1 & im;
        const bool rephit = rc && (uint32_t)(rv >> 8) == (uint32_t)(v1 >> 8);
        bool t0 = valid && mi0 > prefix_idx && ((e0 ^ n0) >> ib) == 0u, t1 = valid && mi1 > prefix_idx && ((e1 ^ n1) >> ib) == 0u;
        const unsigned long long repm = __ballot(rephit);
        const uint32_t nvalid = (uint32_t)__builtin_popcountll(__ballot(valid));

        uint32_t T = 63u, cur0 = 0, mlen = 0, offcode = 0, m = 0, known = 0, blim = 0, fwd = 0, back = 0, r1s = 0, seq_ip = 0;
        uint64_t A = 0, B = 0, vas = 0;
        bool hit = false, isrep = false;
        for (;;) { /* decided again only when a tag lied */
            const unsigned long long m0m = __ballot(t0), m1m = __ballot(t1);
            const unsigned long long hitm = repm | m0m | m1m;
            if (!hitm) break;
            T = ctz64(hitm);
            cur0 = __builtin_amdgcn_readlane(i0, T);
            uint32_t vword;
            if ((repm >> T) & 1ull) {
                const uint32_t rlo = __builtin_amdgcn_readlane((uint32_t)rv, T), vlo = __builtin_amdgcn_readlane((uint32_t)v1, T);
                const uint32_t b = ((rlo ^ vlo) & 0xFFu) == 0u ? 1u : 0u; /* ip2[-1] == repMatch[-1] */
                seq_ip = cur0 + 2u - b;
                m = seq_ip - off1;
                known = 4u + b;
                blim = 0;
                isrep = true;
                vword = 0;
            } else {
                if ((m0m >> T) & 1ull) { seq_ip = cur0; m = __builtin_amdgcn_readlane(mi0, T); vword = __builtin_amdgcn_readlane((uint32_t)v0, T); }
                else { seq_ip = cur0 + 1u; m = __builtin_amdgcn_readlane(mi1, T); vword = __builtin_amdgcn_readlane((uint32_t)v1, T); }
                known = 4u;
                const uint32_t la = seq_ip - anchor, lm = m - prefix_idx;
                blim = la < lm ? la : lm;
                isrep = false;
            }
            /* trip 2: the verification word with the match extension and everything the sequence's tail needs, whatever
             * the match length turns out to be (below 64 bytes more): a window of the input behind the known part of
             * the match -- lane j reads 8 bytes at wb + 2j (A) and wb + 2j + 1 (B), wb = ip + known - 2 -- serves the
             * bytes at the end of the match (ip' - 2 for the complementary insertion, ip' for the immediate-repeat
             * check), the next step's search input and the next literal run; lane j also reads the immediate-repeat
             * candidate for a match ending at ip + known + j (the second offset after this sequence: off1 for a new
             * offset, off2 for a repeat). */
            const uint32_t o2 = isrep ? off2 : off1;
            const uint32_t wb = seq_ip + known - 2u;
            const bool win_a = wb + 2u * lane + 8u <= iend, win_b = wb + 2u * lane + 9u <= iend;
            const uint32_t sver = ld32v(base + m);
            A = ld64v(base + (win_a ? wb + 2u * lane : seq_ip));
            B = ld64v(base + (win_b ? wb + 2u * lane + 1u : seq_ip));
            r1s = ld32v(base + ((o2 > 0u && seq_ip + known + lane + 4u <= iend) ? seq_ip + known + lane - o2 : seq_ip));
            vas = ld64v(base + cur0 + 2u);
            count_both(base + seq_ip + known, base + m + known, base + iend, base + seq_ip, base + m, blim, lane, fwd, back);
            if (isrep || uni(sver) == vword) { hit = true; break; }
            /* the tag lied: strike this candidate out and decide again */
            if (lane == T) { if ((m0m >> T) & 1ull) t0 = false; else t1 = false; }
        }
        {
            const uint32_t ncommit = hit ? T + 1u : nvalid;
            if (lane < ncommit) table[h0] = n0;
            asm volatile("" ::: "memory");
            if (lane < ncommit) table[h1] = n1;
            if (!hit) { ip = (uint32_t)__builtin_amdgcn_readlane(inext, ncommit - 1u); Wc = Wc * 2u < 64u ? Wc * 2u : 64u; continue; }
            Wc = W;
        }
        if (isrep) offcode = 0;
        else { off2 = off1; off1 = seq_ip - m; offcode = off1 + 2u; }
        mlen = known + fwd + back;
        ip = seq_ip - back;
        const uint32_t ll = ip - anchor, seq_anchor = anchor;
        ip += mlen;
        anchor = ip;
        if (fwd < 64u && ip <= ilimit) {
            /* window position t <-> index wb + t; the match ends at t = fwd + 2 */
            auto win64 = [&](uint32_t t) { /* 8 bytes at window position t (per lane), t <= 127 */
                const int j = (int)(t >> 1);
                const uint64_t a = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(A >> 32), j, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)A, j, 64);
                const uint64_t b = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(B >> 32), j, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)B, j, 64);
                return (t & 1u) ? b : a;
            };
            const uint32_t te = fwd + 2u;
            const uint32_t r0 = (uint32_t)uni64(win64(te));
            const uint32_t r1 = __builtin_amdgcn_readlane(r1s, fwd);
            if (!(off2 > 0u && r0 == r1)) {
                const uint64_t va = uni64(vas), vb = uni64(win64(fwd));
                store_seq_pre(ws, ss, ll, litv, base + seq_anchor, offcode, mlen - 3u, lane);
                if (lane == 0) {
                    table[hashs_v(va, hlog, mls)] = ((cur0 + 2u) & im) | ((fast_tag((uint32_t)va) >> ib) << ib);
                    table[hashs_v(vb, hlog, mls)] = ((ip - 2u) & im) | ((fast_tag((uint32_t)vb) >> ib) << ib);
                }
                /* next step: lane k looks at ip + k * step_size (and + 1) */
                const bool nv = lane < W && ip + lane * step_size + 1u < ilimit;
                const uint32_t t0w = te + lane * step_size;
                const bool inw = t0w + 1u <= 127u;
                v0n = win64(inw ? t0w : 0u);
                v1n = win64(inw ? t0w + 1u : 0u);
                if (nv && !inw) { v0n = ld64v(base + ip + lane * step_size); v1n = ld64v(base + ip + lane * step_size + 1u); }
                if (!nv) { v0n = 0ull; v1n = 0ull; }
                /* next literal run: byte at ip + lane = window position te + lane, inside A[j] for 2j <= t <= 2j + 7 */
                {
                    const uint32_t t = te + lane, j = (t >> 1) < 63u ? (t >> 1) : 63u;
                    const uint64_t a = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(A >> 32), (int)j, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)A, (int)j, 64);
                    const bool ok = wb_ok(seq_ip + known - 2u, j, iend); /* lane j's A was a real window read */
                    uint32_t byte = (uint32_t)(a >> (8u * (t - 2u * j))) & 0xFFu;
                    if (!ok && ip + lane < iend) byte = base[ip + lane];
                    litv = ip + lane < iend ? byte : 0u;
                }
                have = true;
                continue;
            }
        }
        bool first = true;
        if (ip > ilimit) store_seq_pre(ws, ss, ll, litv, base + seq_anchor, offcode, mlen - 3u, lane);
        while (ip <= ilimit) {
            const bool nv = lane < W && ip + lane * step_size + 1u < ilimit;
            v0n = nv ? ld64v(base + ip + lane * step_size) : 0ull;
            v1n = nv ? ld64v(base + ip + lane * step_size + 1u) : 0ull;
            const uint32_t litn = ip + lane < iend ? base[ip + lane] : 0u;
            const uint32_t r0v = ld32v(base + ip), r1v = ld32v(base + ip - off2); /* all loads first, see block_dfast_batch */
            const uint64_t vav = ld64v(base + (first ? cur0 + 2u : ip)), vbv = ld64v(base + ip - 2u);
            const uint32_t r0 = uni(r0v), r1 = uni(r1v);
            if (first) {
                const uint64_t va = uni64(vav), vb = uni64(vbv);
                store_seq_pre(ws, ss, ll, litv, base + seq_anchor, offcode, mlen - 3u, lane);
                if (lane == 0) {
                    table[hashs_v(va, hlog, mls)] = ((cur0 + 2u) & im) | ((fast_tag((uint32_t)va) >> ib) << ib);
                    table[hashs_v(vb, hlog, mls)] = ((ip - 2u) & im) | ((fast_tag((uint32_t)vb) >> ib) << ib);
                }
                first = false;
            }
            litv = litn;
            if (!(off2 > 0u && r0 == r1)) { have = true; break; }
            const uint32_t rlen = count_match(base + ip + 4u, base + ip + 4u - off2, base + iend, lane) + 4u;
            const uint32_t t = off2; off2 = off1; off1 = t;
            {
                const uint64_t v = ld64u(base + ip);
                if (lane == 0) table[hashs_v(v, hlog, mls)] = (ip & im) | ((fast_tag((uint32_t)v) >> ib) << ib);
            }
            store_seq_pre(ws, ss, 0, 0, base + anchor, 0, rlen - 3u, lane);
            ip += rlen;
            anchor = ip;
        }
    }
    rep[0] = off1 ? off1 : saved;
    rep[1] = off2 ? off2 : saved;
    return iend - anchor;
}
