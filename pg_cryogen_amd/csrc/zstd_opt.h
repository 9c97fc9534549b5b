/*
 * zstd_opt.h -- the optimal-parser strategies `btopt`, `btultra` and `btultra2` (libzstd 1.4.8 lib/compress/zstd_opt.c:
 * ZSTD_compressBlock_opt_generic over ZSTD_insertBt1 / ZSTD_insertBtAndGetAllMatches; zstd levels 11 .. 22 up to 16 KiB,
 * 13 .. 22 up to 256 KiB, 16 .. 22 above).  Included by zstd_enc.hip inside its namespace, after zstd_lazy.h.
 *
 * Replaces the match-finding half of ZSTD_compress(dst, bound, src, B, level) (reference compression.c:102-104) at the
 * levels whose strategy is an optimal parser; restated for the CPU in oracle/zstd_enc_oracle.c (block_opt,
 * bt_get_all_matches, bt_insert1).
 *
 * The parse is a chain of decisions, each depending on the statistics the previous ones left, so the walk is wave-uniform
 * like the other deep strategies'.  The wave's lanes share the work inside a step: candidate extension 64 bytes at a time,
 * the 3-byte hash table 64 positions at a time, the price of 64 match lengths at a time, the literal histogram.  Per
 * workgroup, behind the hash table and the tree: the 3-byte hash table (minMatch 3), the price table (4096 + 2 positions of
 * 32 bytes), the match ladder, and the statistics between blocks (during a block they live in LDS, over the entropy
 * stage's scratch).
 */
#pragma once

constexpr uint32_t kOptNum = 4096u;
constexpr int kOptMaxPrice = 1 << 30;
constexpr uint32_t kBitCost = 256u;
constexpr uint32_t kOsLit = 0u, kOsLL = 256u, kOsML = kOsLL + kMaxLL + 1u, kOsOF = kOsML + kMaxML + 1u, kOsWords = kOsOF + kMaxOff + 1u;
static_assert(kOsWords * 4u <= kDfMark, "the statistics lie where the other finders keep their mark array");

struct OptT { int price; uint32_t off, mlen, litlen; uint32_t rep[3]; uint32_t pad; };
static_assert(sizeof(OptT) == 32, "two 16-byte halves: the decision, the repeat offsets");
constexpr size_t kOptTabBytes = (size_t)(kOptNum + 2u) * sizeof(OptT);
constexpr size_t kOptMatchBytes = (size_t)(kOptNum + 2u) * 8u;
constexpr size_t kOptExtraBytes = kOptTabBytes + kOptMatchBytes + 2048u; /* + statistics between blocks */

struct OptStats {
    uint32_t *f; /* LDS: lit[256], ll[36], ml[53], of[32] */
    uint32_t lit_sum, ll_sum, ml_sum, of_sum;
    uint32_t lit_base, ll_base, ml_base, of_base;
    bool predef;
};

__device__ inline void lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ inline uint32_t wave_sum32(uint32_t v)
{
    for (int o = 32; o; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return uni(v);
}
__device__ inline uint32_t opt_weight(uint32_t stat, int lvl)
{
    const uint32_t s = stat + 1u, h = hbit(s);
    return lvl ? h * kBitCost + ((s << 8) >> h) : h * kBitCost;
}
__device__ inline void opt_set_base(OptStats &o, int lvl)
{
    o.lit_base = opt_weight(o.lit_sum, lvl);
    o.ll_base = opt_weight(o.ll_sum, lvl);
    o.ml_base = opt_weight(o.ml_sum, lvl);
    o.of_base = opt_weight(o.of_sum, lvl);
}
/* ZSTD_downscaleStat / ZSTD_upscaleStat over t[0 .. last] */
__device__ inline uint32_t opt_downscale(uint32_t *t, uint32_t last, uint32_t shift, uint32_t lane)
{
    uint32_t sum = 0;
    for (uint32_t s = lane; s <= last; s += 64u) { const uint32_t v = 1u + (t[s] >> shift); t[s] = v; sum += v; }
    lds_sync();
    return wave_sum32(sum);
}
__device__ inline uint32_t opt_upscale(uint32_t *t, uint32_t last, uint32_t lane)
{
    uint32_t sum = 0;
    for (uint32_t s = lane; s <= last; s += 64u) { const uint32_t v = (t[s] << 4) - 1u; t[s] = v; sum += v; }
    lds_sync();
    return wave_sum32(sum);
}
/* ZSTD_rescaleFreqs */
__device__ inline void opt_rescale(OptStats &o, const uint8_t *src, uint32_t n, int lvl, uint32_t lane)
{
    o.predef = false;
    if (o.ll_sum == 0u) {
        if (n <= 1024u) o.predef = true;
        for (uint32_t i = lane; i < kOsWords; i += 64u) o.f[i] = i < kOsLL ? 0u : 1u;
        lds_sync();
        for (uint32_t i = lane; i < n; i += 64u) atomicAdd(&o.f[kOsLit + src[i]], 1u);
        lds_sync();
        o.lit_sum = opt_downscale(o.f + kOsLit, 255u, 5u, lane);
        o.ll_sum = kMaxLL + 1u; o.ml_sum = kMaxML + 1u; o.of_sum = kMaxOff + 1u;
    } else {
        o.lit_sum = opt_downscale(o.f + kOsLit, 255u, 5u, lane);
        o.ll_sum = opt_downscale(o.f + kOsLL, kMaxLL, 4u, lane);
        o.ml_sum = opt_downscale(o.f + kOsML, kMaxML, 4u, lane);
        o.of_sum = opt_downscale(o.f + kOsOF, kMaxOff, 4u, lane);
    }
    opt_set_base(o, lvl);
}
__device__ inline uint32_t opt_lit_cost1(const OptStats &o, uint32_t c, int lvl)
{
    if (o.predef) return 6u * kBitCost;
    return o.lit_base - opt_weight(o.f[kOsLit + c], lvl);
}
__device__ inline uint32_t opt_llcode(uint32_t ll) { return ll > 63u ? hbit(ll) + 19u : kLLCode[ll]; }
__device__ inline uint32_t opt_mlcode(uint32_t mb) { return mb > 127u ? hbit(mb) + 36u : kMLCode[mb]; }
__device__ inline uint32_t opt_ll_price(const OptStats &o, uint32_t ll, int lvl)
{
    if (o.predef) return opt_weight(ll, lvl);
    const uint32_t c = opt_llcode(ll);
    return kELLBits[c] * kBitCost + o.ll_base - opt_weight(o.f[kOsLL + c], lvl);
}
/* ZSTD_getMatchPrice (any lane's own mlen) */
__device__ inline uint32_t opt_match_price(const OptStats &o, uint32_t off, uint32_t mlen, int lvl)
{
    const uint32_t oc = hbit(off + 1u), mb = mlen - 3u;
    if (o.predef) return opt_weight(mb, lvl) + (16u + oc) * kBitCost;
    uint32_t price = oc * kBitCost + (o.of_base - opt_weight(o.f[kOsOF + oc], lvl));
    if (lvl < 2 && oc >= 20u) price += (oc - 19u) * 2u * kBitCost; /* long offsets handicapped below btultra */
    const uint32_t mc = opt_mlcode(mb);
    price += kEMLBits[mc] * kBitCost + (o.ml_base - opt_weight(o.f[kOsML + mc], lvl));
    return price + kBitCost / 5u;
}
/* ZSTD_updateStats */
__device__ inline void opt_update_stats(OptStats &o, uint32_t ll, const uint8_t *lit, uint32_t offcode, uint32_t mlen, uint32_t lane)
{
    for (uint32_t u = lane; u < ll; u += 64u) atomicAdd(&o.f[kOsLit + lit[u]], 2u);
    o.lit_sum += ll * 2u;
    if (lane == 0) {
        o.f[kOsLL + opt_llcode(ll)]++;
        o.f[kOsOF + hbit(offcode + 1u)]++;
        o.f[kOsML + opt_mlcode(mlen - 3u)]++;
    }
    o.ll_sum++; o.of_sum++; o.ml_sum++;
    lds_sync();
}

/* ZSTD_insertBt1: one position into the sorted tree; returns how many positions to advance */
__device__ inline uint32_t bt_insert1(HcState &hc, const CPar &cp, const uint8_t *base, uint32_t cur, uint32_t iend, int mls, uint32_t lane)
{
    uint32_t *const bt = hc.chain;
    const uint32_t bt_mask = (1u << (cp.clog - 1)) - 1u;
    const uint32_t h = hashs_v(ld64u(base + cur), cp.hlog, mls);
    uint32_t mi = uni(hc.hash[h]);
    uint32_t common_smaller = 0, common_larger = 0;
    const uint32_t bt_low = bt_mask >= cur ? 0u : cur - bt_mask;
    uint32_t smaller_at = 2u * (cur & bt_mask), larger_at = smaller_at + 1u; /* indexes into bt; 0xFFFFFFFF: nowhere */
    uint32_t match_end = cur + 8u + 1u, best = 8u;
    uint32_t nb = 1u << cp.slog;
    if (lane == 0) hc.hash[h] = cur;
    while (nb-- && mi >= hc.low) {
        const uint32_t next_at = 2u * (mi & bt_mask);
        const uint2 nx = *reinterpret_cast<const uint2 *>(bt + next_at);
        uint32_t ml = common_smaller < common_larger ? common_smaller : common_larger;
        bool m_less;
        ml += bt_count_cmp(base, cur + ml, mi + ml, iend, &m_less, lane);
        if (ml > best) {
            best = ml;
            if (ml > match_end - mi) match_end = mi + ml;
        }
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
    uint32_t positions = 0;
    if (best > 384u) positions = best - 384u < 192u ? best - 384u : 192u;
    const uint32_t fwd = match_end - (cur + 8u);
    return positions > fwd ? positions : fwd;
}

__device__ inline uint32_t hash3_v(uint32_t v, int hlog) { return ((v << 8) * 506832829u) >> (32 - hlog); }

/* ZSTD_BtGetAllMatches + ZSTD_insertBtAndGetAllMatches: the repeat offsets, the 3-byte hash (minMatch 3), then the tree
 * search that inserts the position; matches[] (uint2 {offcode, length}) come out by increasing length */
__device__ uint32_t bt_get_all_matches(uint2 *matches, HcState &hc, const CPar &cp, const uint8_t *base, uint32_t *next3, uint32_t cur,
                                       uint32_t iend, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t ll0, uint32_t length_to_beat,
                                       uint32_t lane)
{
    const int mls = cp.mml;
    const uint32_t min_match = (mls == 3) ? 3u : 4u;
    const uint32_t sufficient = (uint32_t)cp.tlen < kOptNum - 1u ? (uint32_t)cp.tlen : kOptNum - 1u;
    uint32_t *const bt = hc.chain;
    const uint32_t bt_mask = (1u << (cp.clog - 1)) - 1u;
    uint32_t mnum = 0, best = length_to_beat - 1u;
    if (cur < hc.next_to_update) return 0; /* skipped area */
    for (uint32_t idx = hc.next_to_update; idx < cur;) idx += bt_insert1(hc, cp, base, idx, iend, mls, lane);
    hc.next_to_update = cur;

    const uint32_t h = hashs_v(ld64u(base + cur), cp.hlog, mls);
    uint32_t mi = uni(hc.hash[h]);
    uint32_t common_smaller = 0, common_larger = 0;
    const uint32_t bt_low = bt_mask >= cur ? 0u : cur - bt_mask;
    const uint32_t max_dist = 1u << cp.wlog;
    const uint32_t window_low = (cur - hc.low > max_dist) ? cur - max_dist : hc.low;
    const uint32_t match_low = window_low ? window_low : 1u;
    uint32_t smaller_at = 2u * (cur & bt_mask), larger_at = smaller_at + 1u;
    uint32_t match_end = cur + 8u + 1u;
    uint32_t nb = 1u << cp.slog;
    const uint32_t here = ld32u(base + cur);
    /* repeat offsets */
    for (uint32_t rc = ll0; rc < 3u + ll0; rc++) {
        const uint32_t roff = (rc == 3u) ? r0 - 1u : (rc == 0u ? r0 : (rc == 1u ? r1 : r2));
        uint32_t rlen = 0;
        if (roff - 1u < cur - hc.low) { /* discards 0 and anything reaching below the prefix start */
            const uint32_t there = ld32u(base + cur - roff);
            const bool same = min_match == 3u ? (here << 8) == (there << 8) : here == there;
            if (cur - roff >= window_low && same)
                rlen = count_match(base + cur + min_match, base + cur + min_match - roff, base + iend, lane) + min_match;
        }
        if (rlen > best) {
            best = rlen;
            if (lane == 0) matches[mnum] = make_uint2(rc - ll0, rlen);
            mnum++;
            if (rlen > sufficient || cur + rlen == iend) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); return mnum; }
        }
    }
    /* 3-byte matches through their own hash table (ZSTD_insertAndFindFirstIndexHash3: every position up to here goes in,
     * the latest wins -- indexes only grow, so a maximum does it) */
    if (mls == 3 && best < 3u) {
        for (uint32_t idx = *next3; idx < cur; idx += 64u) {
            const uint32_t my = idx + lane;
            if (my < cur) atomicMax(&hc.hash3[hash3_v(ld32v(base + my), hc.hlog3)], my);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        *next3 = cur;
        const uint32_t m3 = uni(hc.hash3[hash3_v(here, hc.hlog3)]);
        if (m3 >= match_low && cur - m3 < (1u << 18)) {
            const uint32_t ml = count_match(base + cur, base + m3, base + iend, lane);
            if (ml >= 3u) {
                best = ml;
                if (lane == 0) matches[0] = make_uint2((cur - m3) + 2u, ml);
                mnum = 1;
                if (ml > sufficient || cur + ml == iend) {
                    hc.next_to_update = cur + 1u; /* skip insertion */
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    return 1;
                }
            }
        }
    }
    if (lane == 0) hc.hash[h] = cur;
    while (nb-- && mi >= match_low) {
        const uint32_t next_at = 2u * (mi & bt_mask);
        const uint2 nx = *reinterpret_cast<const uint2 *>(bt + next_at);
        uint32_t ml = common_smaller < common_larger ? common_smaller : common_larger;
        bool m_less;
        ml += bt_count_cmp(base, cur + ml, mi + ml, iend, &m_less, lane);
        if (ml > best) {
            if (ml > match_end - mi) match_end = mi + ml;
            best = ml;
            if (lane == 0) matches[mnum] = make_uint2((cur - mi) + 2u, ml);
            mnum++;
            if (ml > kOptNum || cur + ml == iend) break; /* dropped, to keep the tree consistent */
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
    hc.next_to_update = match_end - 8u; /* skip repetitive patterns */
    return mnum;
}

/* ZSTD_updateRep */
__device__ inline void opt_update_rep(uint32_t out[3], uint32_t p0, uint32_t p1, uint32_t p2, uint32_t off, uint32_t ll0)
{
    if (off >= 3u) { out[2] = p1; out[1] = p0; out[0] = off - 2u; }
    else {
        const uint32_t rc = off + ll0;
        if (rc > 0u) {
            const uint32_t cur_off = (rc == 3u) ? p0 - 1u : (rc == 1u ? p1 : p2);
            out[2] = (rc >= 2u) ? p1 : p2;
            out[1] = p0;
            out[0] = cur_off;
        } else { out[0] = p0; out[1] = p1; out[2] = p2; }
    }
}

__device__ inline uint4 opt_ld_head(const OptT *opt, uint32_t i) { const uint4 v = *reinterpret_cast<const uint4 *>(opt + i); return make_uint4(uni(v.x), uni(v.y), uni(v.z), uni(v.w)); }
__device__ inline uint4 opt_ld_rep(const OptT *opt, uint32_t i) { const uint4 v = reinterpret_cast<const uint4 *>(opt + i)[1]; return make_uint4(uni(v.x), uni(v.y), uni(v.z), 0u); }

/* ZSTD_compressBlock_opt_generic (oracle: block_opt).  lvl 0: btopt; 2: btultra / btultra2.  Indexes are the library's
 * (base + index = address). */
__device__ uint32_t block_opt(HcState &hc, const CPar &cp, OptStats &o, OptT *opt, uint2 *matches, const uint8_t *base,
                              const uint8_t *istart, uint32_t n, uint32_t *rep, uint8_t *ws, SeqStore &ss, int lvl, uint32_t lane)
{
    uint32_t ip = (uint32_t)(istart - base), anchor = ip;
    const uint32_t iend = ip + n, ilimit = iend - 8u;
    const uint32_t sufficient = (uint32_t)cp.tlen < kOptNum - 1u ? (uint32_t)cp.tlen : kOptNum - 1u;
    const uint32_t min_match = (cp.mml == 3) ? 3u : 4u;
    uint32_t next3 = hc.next_to_update;
    opt_rescale(o, istart, n, lvl, lane);
    if (ip == hc.low) ip++;
    while (ip < ilimit) {
        uint32_t cur, last_pos = 0;
        uint4 last_seq = make_uint4(0, 0, 0, 0); /* {price, off, mlen, litlen} */
        bool shortcut = false;
        {
            const uint32_t litlen = ip - anchor;
            const uint32_t ll0 = litlen == 0u ? 1u : 0u;
            const uint32_t nbm = bt_get_all_matches(matches, hc, cp, base, &next3, ip, iend, rep[0], rep[1], rep[2], ll0, min_match, lane);
            if (!nbm) { ip++; continue; }
            const uint32_t price0 = opt_ll_price(o, litlen, lvl);
            if (lane == 0) {
                *reinterpret_cast<uint4 *>(opt + 0) = make_uint4(price0, 0u, 0u, litlen);
                reinterpret_cast<uint4 *>(opt + 0)[1] = make_uint4(rep[0], rep[1], rep[2], 0u);
            }
            const uint2 top = matches[nbm - 1u];
            const uint32_t max_ml = uni(top.y), max_off = uni(top.x);
            if (max_ml > sufficient) {
                last_seq = make_uint4(0u, max_off, max_ml, litlen);
                cur = 0;
                last_pos = litlen + max_ml;
                shortcut = true;
            } else {
                const uint32_t lit_price = price0 + opt_ll_price(o, 0u, lvl);
                if (lane + 1u < min_match) opt[lane + 1u].price = kOptMaxPrice;
                uint32_t pos = min_match;
                for (uint32_t k = 0; k < nbm; k++) {
                    const uint2 m = matches[k];
                    const uint32_t off = uni(m.x), end = uni(m.y);
                    for (uint32_t p = pos + lane; p <= end; p += 64u)
                        *reinterpret_cast<uint4 *>(opt + p) = make_uint4(lit_price + opt_match_price(o, off, p, lvl), off, p, litlen);
                    pos = end + 1u;
                }
                last_pos = pos - 1u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        }
        if (!shortcut) {
            /* opt[cur - 1] as decided, carried in registers */
            uint4 prev = opt_ld_head(opt, 0);
            uint32_t pr0 = rep[0], pr1 = rep[1], pr2 = rep[2];
            for (cur = 1; cur <= last_pos; cur++) {
                const uint32_t inr = ip + cur;
                uint4 me = opt_ld_head(opt, cur);
                {
                    const uint32_t litlen = (prev.z == 0u) ? prev.w + 1u : 1u;
                    const int price = (int)prev.x + (int)opt_lit_cost1(o, uni((uint32_t)base[inr - 1u]), lvl) + (int)opt_ll_price(o, litlen, lvl)
                                      - (int)opt_ll_price(o, litlen - 1u, lvl);
                    if (price <= (int)me.x) {
                        me = make_uint4((uint32_t)price, 0u, 0u, litlen);
                        if (lane == 0) *reinterpret_cast<uint4 *>(opt + cur) = me;
                    }
                }
                uint32_t mr[3];
                if (me.z != 0u) {
                    const uint4 pr = opt_ld_rep(opt, cur - me.z);
                    opt_update_rep(mr, pr.x, pr.y, pr.z, me.y, me.w == 0u ? 1u : 0u);
                } else { mr[0] = pr0; mr[1] = pr1; mr[2] = pr2; }
                if (lane == 0) reinterpret_cast<uint4 *>(opt + cur)[1] = make_uint4(mr[0], mr[1], mr[2], 0u);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                prev = me; pr0 = mr[0]; pr1 = mr[1]; pr2 = mr[2];
                if (inr > ilimit) continue; /* the last match starts at least 8 bytes before the end */
                if (cur == last_pos) break;
                if (lvl == 0 && uni((uint32_t)opt[cur + 1u].price) <= me.x + kBitCost / 2u) continue; /* btopt skips unpromising positions */
                const uint32_t ll0 = me.z != 0u ? 1u : 0u;
                const uint32_t litlen = me.z == 0u ? me.w : 0u;
                const uint32_t base_price = me.x + opt_ll_price(o, 0u, lvl);
                const uint32_t nbm = bt_get_all_matches(matches, hc, cp, base, &next3, inr, iend, mr[0], mr[1], mr[2], ll0, min_match, lane);
                if (!nbm) continue;
                {
                    const uint2 top = matches[nbm - 1u];
                    const uint32_t max_ml = uni(top.y);
                    if (max_ml > sufficient || cur + max_ml >= kOptNum) {
                        last_seq = make_uint4(0u, uni(top.x), max_ml, litlen);
                        cur -= (me.z == 0u) ? me.w : 0u; /* may wrap: then it is the first sequence */
                        last_pos = cur + litlen + max_ml;
                        if (cur > kOptNum) cur = 0;
                        shortcut = true;
                        break;
                    }
                }
                /* positions between the old end and the shortest match: no price yet */
                for (uint32_t p = last_pos + 1u + lane; p < cur + min_match; p += 64u) opt[p].price = kOptMaxPrice;
                uint32_t new_last = last_pos;
                for (uint32_t k = 0; k < nbm; k++) {
                    const uint2 m = matches[k];
                    const uint32_t off = uni(m.x), last_ml = uni(m.y);
                    const uint32_t start_ml = k > 0 ? uni(matches[k - 1u].y) + 1u : min_match;
                    /* scan downward, 64 lengths per step; btopt stops a match at the first length that does not improve */
                    for (uint32_t hi = last_ml;; hi -= 64u) {
                        const uint32_t mlen = hi - lane;
                        const bool in = hi >= start_ml + lane; /* mlen >= start_ml, without wrapping */
                        const uint32_t pos = cur + mlen;
                        int price = 0;
                        bool better = false;
                        if (in) {
                            price = (int)(base_price + opt_match_price(o, off, mlen, lvl));
                            better = pos > last_pos || price < opt[pos].price;
                        }
                        unsigned long long worse = __ballot(in && !better);
                        bool stop = false;
                        if (lvl == 0 && worse != 0ull) { /* lanes below the first one that fails */
                            const uint32_t first = ctz64(worse);
                            better = better && lane < first;
                            stop = true;
                        }
                        if (better) *reinterpret_cast<uint4 *>(opt + pos) = make_uint4((uint32_t)price, off, mlen, litlen);
                        if (stop || hi < start_ml + 64u) break;
                    }
                    if (cur + last_ml > new_last) new_last = cur + last_ml;
                }
                last_pos = new_last;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
            if (!shortcut) {
                last_seq = opt_ld_head(opt, last_pos);
                cur = last_pos > last_seq.w + last_seq.z ? last_pos - (last_seq.w + last_seq.z) : 0u;
            }
        }
        /* shortest path: the next stretch's repeat offsets, then the chosen sequences, backwards into place */
        {
            const uint4 cr = opt_ld_rep(opt, cur);
            if (last_seq.z != 0u) opt_update_rep(rep, cr.x, cr.y, cr.z, last_seq.y, last_seq.w == 0u ? 1u : 0u);
            else { rep[0] = cr.x; rep[1] = cr.y; rep[2] = cr.z; }
        }
        {
            const uint32_t store_end = cur + 1u;
            uint32_t store_start = store_end, seq_pos = cur;
            if (lane == 0) *reinterpret_cast<uint4 *>(opt + store_end) = last_seq;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            while (seq_pos > 0u) {
                const uint4 e = opt_ld_head(opt, seq_pos);
                const uint32_t back = e.w + e.z;
                store_start--;
                if (lane == 0) *reinterpret_cast<uint4 *>(opt + store_start) = e;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                seq_pos = (seq_pos > back) ? seq_pos - back : 0u;
            }
            for (uint32_t sp = store_start; sp <= store_end; sp++) {
                const uint4 e = opt_ld_head(opt, sp);
                const uint32_t llen = e.w, mlen = e.z, offcode = e.y;
                if (mlen == 0u) { ip = anchor + llen; continue; } /* only literals: the last entry, starts the next stretch */
                opt_update_stats(o, llen, base + anchor, offcode, mlen, lane);
                store_seq(ws, ss, llen, base + anchor, offcode, mlen - 3u, lane);
                anchor += llen + mlen;
                ip = anchor;
            }
            opt_set_base(o, lvl);
        }
    }
    return iend - anchor;
}
