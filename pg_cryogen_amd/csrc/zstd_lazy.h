/*
 * zstd_lazy.h -- the `greedy`, `lazy`, `lazy2` and `btlazy2` strategies' match finder (libzstd 1.4.8
 * ZSTD_compressBlock_lazy_generic at depth 0, 1, 2 over the hash-chain searcher ZSTD_HcFindBestMatch, and at depth 2 over
 * the binary-tree searcher ZSTD_BtFindBestMatch, no dictionary; zstd levels 5 .. 12 at cryo block sizes, .. 15 above 256 KiB).  Included by zstd_enc.hip inside its namespace, after zstd_dfast.h.
 *
 * Replaces the match-finding half of ZSTD_compress(dst, bound, src, B, level) (reference
 * compression.c:102-104); restated for the CPU in oracle/zstd_enc_oracle.c (block_lazy, hc_find_best).
 *
 * First correct version: the walk is wave-uniform.  Every position up to the one searched is inserted into the
 * hash table and the chain table (64 positions per step as long as their slots differ -- the mark array of the
 * other finders -- which is exact: chain[idx] = hash[h]; hash[h] = idx for distinct h in any order); a search
 * walks at most 2^searchLog chain links, each a dependent load, and extends candidates 64 bytes per step.
 * Tables: hash (2^hashLog u32) then chain (2^chainLog u32) behind the workgroup's workspace.  `lazy` / `lazy2`
 * (levels 6 .. 10) run the same searcher again at ip+1 (and ip+2) and choose their sequence tables by estimated cost
 * (zstd_enc.hip select_type).
 */
#pragma once

struct HcState { uint32_t *hash, *chain; uint32_t next_to_update;
                 uint32_t *hash3; int hlog3; uint32_t low; /* zstd_opt.h: 3-byte hash table, lowest valid index */ };

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

/* ZSTD_insertAndFindFirstIndex: positions next_to_update .. target-1 go into the tables; returns the head of
 * target's chain */
__device__ inline uint32_t hc_insert_find(HcState &hc, uint8_t *mark, const CPar &cp, const uint8_t *base, uint32_t target, int mls,
                                          uint32_t lane)
{
    const uint32_t cmask = (1u << cp.clog) - 1u;
    uint32_t idx = hc.next_to_update;
    while (idx < target) {
        const uint32_t my = idx + lane;
        bool on = my < target;
        const uint32_t h = hashs_v(on ? ld64v(base + my) : 0ull, cp.hlog, mls);
        const uint32_t keep = distinct_prefix(mark, h & (kDfMark - 1u), on, lane);
        on = on && lane < keep;
        if (on) { const uint32_t old = hc.hash[h]; hc.chain[my & cmask] = old; hc.hash[h] = my; }
        idx += (uint32_t)__builtin_popcountll(__ballot(on));
    }
    hc.next_to_update = target;
    return uni(hc.hash[hashs_v(ld64u(base + target), cp.hlog, mls)]);
}

/* ZSTD_HcFindBestMatch: longest match among at most 2^searchLog chain links; *offset_ptr = distance + 2.
 * One trip to memory per link: the next link and the candidate's first 64 bytes are requested together (the
 * library's byte test at [ml] before counting is only a shortcut: a candidate that fails it cannot be longer). */
__device__ inline uint32_t hc_find_best(HcState &hc, uint8_t *mark, const CPar &cp, const uint8_t *base, uint32_t cur, uint32_t iend,
                                        uint32_t *offset_ptr, int mls, uint32_t lane)
{
    const uint32_t csize = 1u << cp.clog, cmask = csize - 1u;
    const uint32_t max_dist = 1u << cp.wlog;
    const uint32_t low_limit = (cur - 1u > max_dist) ? cur - max_dist : 1u; /* window.lowLimit = 1 */
    const uint32_t min_chain = cur > csize ? cur - csize : 0u;
    uint32_t attempts = 1u << cp.slog;
    uint32_t ml = 4u - 1u;
    uint32_t mi = hc_insert_find(hc, mark, cp, base, cur, mls, lane);
    const bool inb = cur + lane < iend;
    const uint32_t mine = inb ? base[cur + lane] : 0u; /* the 64 bytes at the search position, one per lane */
    for (; mi >= low_limit && attempts > 0u; attempts--) {
        const uint32_t nxt = hc.chain[mi & cmask];
        const uint32_t theirs = inb ? base[mi + lane] : 256u;
        const unsigned long long neq = __ballot(theirs != mine);
        uint32_t cml = neq ? ctz64(neq) : 64u;
        if (cml == 64u) cml += count_match(base + cur + 64u, base + mi + 64u, base + iend, lane);
        if (cml > ml) {
            ml = cml;
            *offset_ptr = cur - mi + 2u;
            if (cur + cml == iend) break;
        }
        if (mi <= min_chain) break;
        mi = uni(nxt);
    }
    return ml;
}

/* After a search that found nothing, the parser moves on one position and searches again; in a literal run that is a chain
 * of searches that all find nothing.  This takes the next positions of such a run at once, one per lane, and returns how
 * many of them, from `ip` on, are *proven* to find nothing -- the parser then skips them and runs its unchanged search at
 * the first position that may find something, so the result is the serial walk's.  A position finds nothing when the repeat
 * offset does not match at ip+1 and no link of its chain (as many as the serial search would visit) starts with the
 * position's four bytes.  A lane must see the tables as its serial search would: every earlier position inserted.  Those
 * below `ip` are inserted first; the batch's own positions are kept out of each other's chains by cutting the batch in
 * front of the first lane whose hash slot an earlier lane shares (the mark array), and a chain slot an earlier lane would
 * overwrite is older than the chain table's reach, where the walk stops anyway.  Only while the step is one position
 * (the first 256 bytes of a literal run). */
constexpr uint32_t kSkipWidth = 32u;
__device__ inline uint32_t hc_skip_misses(HcState &hc, uint8_t *mark, const CPar &cp, const uint8_t *base, uint32_t ip, uint32_t anchor,
                                          uint32_t ilimit, uint32_t off1, int mls, uint32_t lane)
{
    if ((ip - anchor) >> 8) return 0u;
    const uint32_t csize = 1u << cp.clog, cmask = csize - 1u;
    const uint32_t max_dist = 1u << cp.wlog;
    const uint32_t p = ip + lane;
    bool on = lane < kSkipWidth && p < ilimit && (p - anchor) < 256u;
    (void)hc_insert_find(hc, mark, cp, base, ip, mls, lane);
    const uint64_t v8 = on ? ld64v(base + p) : 0ull;
    const uint32_t h = hashs_v(v8, cp.hlog, mls);
    const uint32_t keep = distinct_prefix(mark, h & (kDfMark - 1u), on, lane);
    on = on && lane < keep;
    const uint32_t here = (uint32_t)v8;
    bool hit = on && off1 > 0u && ld32v(base + (on ? p + 1u - off1 : ip)) == (uint32_t)(v8 >> 8);
    uint32_t mi = on ? hc.hash[h] : 0u;
    const uint32_t low_limit = (p - 1u > max_dist) ? p - max_dist : 1u;
    const uint32_t min_chain = p > csize ? p - csize : 0u;
    uint32_t attempts = 1u << cp.slog;
    bool walking = on && !hit;
    for (;;) {
        const bool active = walking && mi >= low_limit && attempts > 0u;
        if (!__any(active)) break;
        const uint32_t nxt = active ? hc.chain[mi & cmask] : 0u;
        const uint32_t theirs = ld32v(base + (active ? mi : ip));
        if (active) {
            if (theirs == here) { hit = true; walking = false; }   /* at least 8 bytes are left: four equal bytes are a match */
            else if (mi <= min_chain) walking = false;
            else { mi = nxt; attempts--; }
        } else
            walking = false;
    }
    const unsigned long long stop = __ballot(!(on && !hit));
    const uint32_t n = stop ? ctz64(stop) : 64u;
    if (n > 1u) (void)hc_insert_find(hc, mark, cp, base, ip + n - 1u, mls, lane); /* what the last skipped search would have inserted */
    return n;
}

/* ---- strategy `btlazy2`: the lazy2 parser over the binary-tree searcher (libzstd 1.4.8 ZSTD_updateDUBT, ZSTD_insertDUBT1,
 * ZSTD_DUBT_findBestMatch, ZSTD_BtFindBestMatch; oracle: dubt_update, dubt_insert1, bt_find_best).  The chain table is a
 * binary tree of 2^(chainLog-1) nodes, two links per position; positions are first chained unsorted (second link = the
 * mark 1) and sorted into the tree in batches when a search runs into them.  Wave-uniform like the hash-chain searcher:
 * every tree step is a dependent load of a node and a 64-bytes-per-step comparison. */
constexpr uint32_t kDubtUnsorted = 1u;

__device__ inline void dubt_update(HcState &hc, uint8_t *mark, const CPar &cp, const uint8_t *base, uint32_t target, int mls, uint32_t lane)
{
    const uint32_t bt_mask = (1u << (cp.clog - 1)) - 1u;
    uint32_t idx = hc.next_to_update;
    while (idx < target) {
        const uint32_t my = idx + lane;
        bool on = my < target;
        const uint32_t h = hashs_v(on ? ld64v(base + my) : 0ull, cp.hlog, mls);
        const uint32_t keep = distinct_prefix(mark, h & (kDfMark - 1u), on, lane);
        on = on && lane < keep;
        if (on) {
            const uint32_t old = hc.hash[h];
            hc.hash[h] = my;
            *reinterpret_cast<uint2 *>(hc.chain + 2u * (my & bt_mask)) = make_uint2(old, kDubtUnsorted);
        }
        idx += (uint32_t)__builtin_popcountll(__ballot(on));
    }
    hc.next_to_update = target;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}

/* a[0 ..) against b[0 ..), at most up to `end` (an index like cur): common length, 64 bytes per step */
__device__ inline uint32_t bt_count(const uint8_t *base, uint32_t a, uint32_t b, uint32_t iend, uint32_t lane)
{
    return count_match(base + a, base + b, base + iend, lane);
}
/* the same, and which side is smaller at the first difference (*b_less: b's byte < a's byte), taken from the bytes the
 * comparison already holds instead of a second trip to memory; meaningless when the count ran into `iend` */
__device__ inline uint32_t bt_count_cmp(const uint8_t *base, uint32_t a, uint32_t b, uint32_t iend, bool *b_less, uint32_t lane)
{
    const uint8_t *pa = base + a, *pb = base + b, *end = base + iend;
    uint32_t done = 0;
    for (;;) {
        const bool inb = pa + done + lane < end;
        const uint32_t va = inb ? pa[done + lane] : 0u, vb = inb ? pb[done + lane] : 0u;
        const unsigned long long neq = __ballot(!(inb && va == vb));
        if (neq != 0ull) {
            const uint32_t l = ctz64(neq);
            const uint32_t pair = (uint32_t)__builtin_amdgcn_readlane((int)(va | (vb << 8)), (int)l);
            *b_less = (pair >> 8) < (pair & 0xFFu);
            return done + l;
        }
        done += 64u;
    }
}

__device__ inline void dubt_insert1(HcState &hc, const CPar &cp, const uint8_t *base, uint32_t cur, uint32_t iend, uint32_t nb_compares,
                                    uint32_t bt_low, uint32_t lane)
{
    uint32_t *const bt = hc.chain;
    const uint32_t bt_mask = (1u << (cp.clog - 1)) - 1u;
    uint32_t common_smaller = 0, common_larger = 0;
    uint32_t smaller_at = 2u * (cur & bt_mask), larger_at = smaller_at + 1u; /* indexes into bt; 0xFFFFFFFF: nowhere */
    uint32_t mi = uni(bt[smaller_at]);
    const uint32_t max_dist = 1u << cp.wlog;
    const uint32_t window_low = (cur - 1u > max_dist) ? cur - max_dist : 1u;
    while (nb_compares-- && mi > window_low) {
        const uint32_t next_at = 2u * (mi & bt_mask);
        const uint2 nx = *reinterpret_cast<const uint2 *>(bt + next_at);
        uint32_t ml = common_smaller < common_larger ? common_smaller : common_larger;
        bool m_less;
        ml += bt_count_cmp(base, cur + ml, mi + ml, iend, &m_less, lane);
        if (cur + ml == iend) break; /* equal: dropped */
        if (m_less) {
            if (lane == 0 && smaller_at != 0xFFFFFFFFu) bt[smaller_at] = mi;
            common_smaller = ml;
            if (mi <= bt_low) { smaller_at = 0xFFFFFFFFu; break; }
            smaller_at = next_at + 1u;
            mi = uni(nx.y);
        } else {
            if (lane == 0 && larger_at != 0xFFFFFFFFu) bt[larger_at] = mi;
            common_larger = ml;
            if (mi <= bt_low) { larger_at = 0xFFFFFFFFu; break; }
            larger_at = next_at;
            mi = uni(nx.x);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    if (lane == 0) {
        if (smaller_at != 0xFFFFFFFFu) bt[smaller_at] = 0;
        if (larger_at != 0xFFFFFFFFu) bt[larger_at] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}

__device__ inline uint32_t bt_find_best(HcState &hc, uint8_t *mark, const CPar &cp, const uint8_t *base, uint32_t cur, uint32_t iend,
                                        uint32_t *offset_ptr, int mls, uint32_t lane)
{
    uint32_t *const bt = hc.chain;
    const uint32_t bt_mask = (1u << (cp.clog - 1)) - 1u;
    if (cur < hc.next_to_update) return 0; /* skipped area */
    dubt_update(hc, mark, cp, base, cur, mls, lane);
    const uint32_t h = hashs_v(ld64u(base + cur), cp.hlog, mls);
    uint32_t mi = uni(hc.hash[h]);
    const uint32_t max_dist = 1u << cp.wlog;
    const uint32_t window_low = (cur - 1u > max_dist) ? cur - max_dist : 1u;
    const uint32_t bt_low = (bt_mask >= cur) ? 0u : cur - bt_mask;
    const uint32_t unsort_limit = bt_low > window_low ? bt_low : window_low;
    uint32_t nb_compares = 1u << cp.slog, nb_candidates = nb_compares, previous = 0;
    /* reach the end of the unsorted candidates (their marks become a reversed chain to come back by) */
    uint32_t cand_at = 2u * (mi & bt_mask);
    uint2 node = *reinterpret_cast<const uint2 *>(bt + cand_at);
    while (mi > unsort_limit && uni(node.y) == kDubtUnsorted && nb_candidates > 1u) {
        if (lane == 0) bt[cand_at + 1u] = previous;
        previous = mi;
        mi = uni(node.x);
        cand_at = 2u * (mi & bt_mask);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        node = *reinterpret_cast<const uint2 *>(bt + cand_at);
        nb_candidates--;
    }
    /* the last candidate, if still unsorted, is dropped */
    if (mi > unsort_limit && uni(node.y) == kDubtUnsorted) {
        if (lane == 0) *reinterpret_cast<uint2 *>(bt + cand_at) = make_uint2(0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    /* batch sort of the stacked candidates */
    mi = previous;
    while (mi) {
        const uint32_t next_idx = uni(bt[2u * (mi & bt_mask) + 1u]);
        dubt_insert1(hc, cp, base, mi, iend, nb_candidates, unsort_limit, lane);
        mi = next_idx;
        nb_candidates++;
    }
    /* the longest match, inserting the current position on the way */
    uint32_t common_smaller = 0, common_larger = 0, best = 0;
    uint32_t smaller_at = 2u * (cur & bt_mask), larger_at = smaller_at + 1u;
    uint32_t match_end_idx = cur + 8u + 1u;
    mi = uni(hc.hash[h]);
    if (lane == 0) hc.hash[h] = cur;
    while (nb_compares-- && mi > window_low) {
        const uint32_t next_at = 2u * (mi & bt_mask);
        const uint2 nx = *reinterpret_cast<const uint2 *>(bt + next_at);
        uint32_t ml = common_smaller < common_larger ? common_smaller : common_larger;
        bool m_less;
        ml += bt_count_cmp(base, cur + ml, mi + ml, iend, &m_less, lane);
        if (ml > best) {
            if (ml > match_end_idx - mi) match_end_idx = mi + ml;
            if ((4 * (int)(ml - best)) > (int)(hbit(cur - mi + 1u) - hbit(*offset_ptr + 1u))) {
                best = ml;
                *offset_ptr = 2u + cur - mi;
            }
            if (cur + ml == iend) break; /* equal: dropped, to keep the tree consistent */
        }
        if (m_less) {
            if (lane == 0 && smaller_at != 0xFFFFFFFFu) bt[smaller_at] = mi;
            common_smaller = ml;
            if (mi <= bt_low) { smaller_at = 0xFFFFFFFFu; break; }
            smaller_at = next_at + 1u;
            mi = uni(nx.y);
        } else {
            if (lane == 0 && larger_at != 0xFFFFFFFFu) bt[larger_at] = mi;
            common_larger = ml;
            if (mi <= bt_low) { larger_at = 0xFFFFFFFFu; break; }
            larger_at = next_at;
            mi = uni(nx.x);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    if (lane == 0) {
        if (smaller_at != 0xFFFFFFFFu) bt[smaller_at] = 0;
        if (larger_at != 0xFFFFFFFFu) bt[larger_at] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    hc.next_to_update = match_end_idx - 8u; /* skip repetitive patterns */
    return best;
}

/* ZSTD_compressBlock_lazy_generic: depth 0 greedy, 1 lazy, 2 lazy2 (oracle: block_lazy).  Indexes are the library's
 * (base = src - 1).  The walk is wave-uniform; depth 1 / 2 search again at ip+1 (ip+2) and keep the candidate whose
 * gain estimate (4 x length - log2(offset code)) is better. */
__device__ uint32_t block_lazy(HcState &hc, uint8_t *mark, const CPar &cp, int depth, const uint8_t *base, const uint8_t *istart, uint32_t n,
                               uint32_t *rep, uint8_t *ws, SeqStore &ss, uint32_t lane)
{
    const int mls = cp.mml < 4 ? 4 : (cp.mml > 6 ? 6 : cp.mml);
    uint32_t ip = (uint32_t)(istart - base), anchor = ip;
    const uint32_t iend = ip + n, ilimit = iend - 8u;
    const uint32_t prefix_lowest = 1u; /* window.dictLimit: the catch-up is not window-limited */
    uint32_t off1 = rep[0], off2 = rep[1], saved = 0;
    if (ip == prefix_lowest) ip++;
    {
        const uint32_t max_dist = 1u << cp.wlog;
        const uint32_t wlow = (ip - 1u > max_dist) ? ip - max_dist : 1u;
        const uint32_t max_rep = ip - wlow;
        if (off2 > max_rep) { saved = off2; off2 = 0; }
        if (off1 > max_rep) { saved = off1; off1 = 0; }
    }
    while (ip < ilimit) {
        uint32_t mlen = 0, offset = 0, start = ip + 1u;
        bool have = false;
        if (off1 > 0u && ld32u(base + ip + 1u - off1) == ld32u(base + ip + 1u)) {
            mlen = count_match(base + ip + 1u + 4u, base + ip + 1u + 4u - off1, base + iend, lane) + 4u;
            have = depth == 0; /* greedy: taken as it is */
        }
        if (!have) {
            {
                uint32_t off_found = 999999999u;
                const uint32_t ml2 = cp.bt ? bt_find_best(hc, mark, cp, base, ip, iend, &off_found, mls, lane) : hc_find_best(hc, mark, cp, base, ip, iend, &off_found, mls, lane);
                if (ml2 > mlen) { mlen = ml2; start = ip; offset = off_found; }
            }
            if (mlen < 4u) {
                ip += ((ip - anchor) >> 8) + 1u;
                if (!cp.bt && ip < ilimit) ip += hc_skip_misses(hc, mark, cp, base, ip, anchor, ilimit, off1, mls, lane);
                continue;
            }
            if (depth >= 1)
                while (ip < ilimit) {
                    ip++;
                    if (offset && off1 > 0u && ld32u(base + ip) == ld32u(base + ip - off1)) {
                        const uint32_t ml_rep = count_match(base + ip + 4u, base + ip + 4u - off1, base + iend, lane) + 4u;
                        const int gain2 = (int)(ml_rep * 3u);
                        const int gain1 = (int)(mlen * 3u - hbit(offset + 1u) + 1u);
                        if (ml_rep >= 4u && gain2 > gain1) { mlen = ml_rep; offset = 0; start = ip; }
                    }
                    {
                        uint32_t off2f = 999999999u;
                        const uint32_t ml2 = cp.bt ? bt_find_best(hc, mark, cp, base, ip, iend, &off2f, mls, lane) : hc_find_best(hc, mark, cp, base, ip, iend, &off2f, mls, lane);
                        const int gain2 = (int)(ml2 * 4u - hbit(off2f + 1u));
                        const int gain1 = (int)(mlen * 4u - hbit(offset + 1u) + 4u);
                        if (ml2 >= 4u && gain2 > gain1) { mlen = ml2; offset = off2f; start = ip; continue; }
                    }
                    if (depth == 2 && ip < ilimit) {
                        ip++;
                        if (offset && off1 > 0u && ld32u(base + ip) == ld32u(base + ip - off1)) {
                            const uint32_t ml_rep = count_match(base + ip + 4u, base + ip + 4u - off1, base + iend, lane) + 4u;
                            const int gain2 = (int)(ml_rep * 4u);
                            const int gain1 = (int)(mlen * 4u - hbit(offset + 1u) + 1u);
                            if (ml_rep >= 4u && gain2 > gain1) { mlen = ml_rep; offset = 0; start = ip; }
                        }
                        {
                            uint32_t off2f = 999999999u;
                            const uint32_t ml2 = cp.bt ? bt_find_best(hc, mark, cp, base, ip, iend, &off2f, mls, lane) : hc_find_best(hc, mark, cp, base, ip, iend, &off2f, mls, lane);
                            const int gain2 = (int)(ml2 * 4u - hbit(off2f + 1u));
                            const int gain1 = (int)(mlen * 4u - hbit(offset + 1u) + 7u);
                            if (ml2 >= 4u && gain2 > gain1) { mlen = ml2; offset = off2f; start = ip; continue; }
                        }
                    }
                    break;
                }
            if (offset) { /* catch up */
                const uint32_t m = start - (offset - 2u);
                const uint32_t la = start - anchor, lm = m - prefix_lowest;
                const uint32_t back = count_back(base + start, base + m, la < lm ? la : lm, lane);
                start -= back;
                mlen += back;
                off2 = off1;
                off1 = offset - 2u;
            }
        }
        store_seq(ws, ss, start - anchor, base + anchor, offset, mlen - 3u, lane);
        anchor = ip = start + mlen;
        while (ip <= ilimit && off2 > 0u && ld32u(base + ip) == ld32u(base + ip - off2)) {
            const uint32_t rlen = count_match(base + ip + 4u, base + ip + 4u - off2, base + iend, lane) + 4u;
            const uint32_t t = off2; off2 = off1; off1 = t;
            store_seq(ws, ss, 0, base + anchor, 0, rlen - 3u, lane);
            ip += rlen;
            anchor = ip;
        }
    }
    rep[0] = off1 ? off1 : saved;
    rep[1] = off2 ? off2 : saved;
    return iend - anchor;
}
