/*
 * zstd_enc.hip -- Zstandard frame encode, one wavefront per cryo block, output bytes identical
 * to libzstd 1.4.8 at every level (-5 .. 22: the strategies `fast`, `dfast`, `greedy`, `lazy`, `lazy2`, `btlazy2`, and the
 * optimal parsers `btopt`, `btultra`, `btultra2` of zstd_opt.h; the reference's default zstd_compression_level_guc is 1).
 *
 * Replaces ZSTD_compress(dst, ZSTD_compressBound(B), src, B, level)
 * (reference compression.c:102-104).
 *
 * Pipeline per 128 KiB zstd block (a 1 MiB cryo block is a frame of 8 dependent blocks):
 *   match finder  : strategy `fast` (levels -5..2) or `dfast` (3, 4), zstd_dfast.h (and `greedy` / `lazy` /
 *                   `lazy2` / `btlazy2`, levels 5..15, zstd_lazy.h, and the optimal parsers above, zstd_opt.h:
 *                   wave-uniform walks over a hash chain or a binary tree): the walk is a serial
 *                   recurrence over hash tables that do not fit LDS next to the entropy stage, so the
 *                   tables live in global memory and a step takes the next 16-32 search positions at
 *                   once, one per lane, paying the trips to memory once per step (`fast`, round 6: table
 *                   slots + repeat candidates, then verification + match extension + tail -- an entry
 *                   carries a tag of its position's four bytes, so candidates are compared without being
 *                   read; `dfast`: slots, candidates, extension, tail); a serial restatement (block_fast)
 *                   is kept as a testing aid.
 *   literals      : histogram by LDS atomics (all lanes), length-limited Huffman tree
 *                   (serial, <= 256 symbols), weights FSE-compressed or raw, then the 4
 *                   backward bitstreams: all 64 lanes per stream (runs of symbols, bit offsets by scan).
 *   sequences     : codes + histograms lane-parallel; encoding-type choice (thresholds below `lazy`,
 *                   estimated costs + repeat of the previous block's table from `lazy` on), FSE
 *                   normalisation, table description serial; the interleaved LL/OF/ML bitstream reads 64 sequences
 *                   at a time into lanes.
 *   block         : raw fallback when the gain is below size/64 + 2, RLE block for constant
 *                   non-first blocks, repeat offsets / Huffman table state carried over only
 *                   by blocks emitted compressed.
 * Sequences, literals, codes and the finder's tables live in a per-workgroup global workspace.
 */
#include "lz_common.h"
#include <cstddef>
#include <cstdio>
#include <cstdlib>

namespace cryo {

namespace {

constexpr uint32_t kZBlk = 128u << 10;
constexpr uint32_t kMaxSeq = kZBlk / 3u + 8u;
constexpr uint32_t kMaxLL = 35, kMaxML = 52, kMaxOff = 31, kDefMaxOff = 28;
constexpr int kHufLogMaxE = 12;

/* per-workgroup global workspace layout */
constexpr size_t kWsSeq = 0;                                   /* uint2 {off, ll | ml<<16} x kMaxSeq */
constexpr size_t kWsLit = kWsSeq + (size_t)kMaxSeq * 8u;        /* literals                          */
constexpr size_t kWsLlc = kWsLit + kZBlk + 64u;                 /* LL / OF / ML codes                */
constexpr size_t kWsOfc = kWsLlc + kMaxSeq;
constexpr size_t kWsMlc = kWsOfc + kMaxSeq;

struct FseCt {
    uint16_t state[512];
    int32_t dfind[64];
    uint32_t dnb[64];
    int log;
    uint32_t max_sym;          /* maxSymbolValue of the table (the repeat-mode cost test) */
};
constexpr size_t kWsTabs = (kWsMlc + kMaxSeq + 15u) & ~(size_t)15u; /* LL, OF, ML tables of the previous compressed block (`lazy`+) */
constexpr size_t kWsBytes = (kWsTabs + 3u * sizeof(FseCt) + 255u) & ~(size_t)255u; /* the match finder's u32 table(s) follow */

struct EncLds {
    uint32_t hist[256];
    union { /* the literals stage is over before the sequence tables are built */
        struct {
            /* Huffman tree nodes; huffNode[i] of the library lives at index i+1, its "fake entry" at 0 */
            uint32_t ncount[516];
            uint16_t nparent[516];
            uint8_t nbyte[516];
            uint8_t nnb[516];
        };
        struct { FseCt of, ml; };
    };
    uint16_t hval[256];        /* code table built for this block */
    uint8_t hnb[256];
    uint16_t pval[256];        /* table confirmed by the previous compressed block ("check" mode) */
    uint8_t pnb[256];
    uint8_t wts[256];
    int16_t norm[64];
    uint32_t cumul[66];
    uint8_t cell[512];
    uint32_t seg_bits[4];
    FseCt ll;                  /* doubles as the table of the Huffman weights (literals come before sequences) */
};

__device__ inline uint32_t hbit(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }
__device__ inline uint32_t ld32u(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return uni(v); }
__device__ inline uint64_t ld64u(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return uni64(v); }

/* forward LSB-first bit writer into global memory (one lane's view) */
struct BitW {
    uint8_t *p;
    uint64_t acc;
    uint32_t n;
    uint32_t len;
    uint32_t cap; /* bytes that may be written; `len` keeps counting beyond (the caller then drops the result) */
    __device__ inline void init(uint8_t *dst, uint32_t capacity = 0xFFFFFFFFu) { p = dst; acc = 0; n = 0; len = 0; cap = capacity; }
    __device__ inline void add(uint64_t v, uint32_t nb)
    {
        if (nb == 0u) return;
        acc |= (v & ((1ull << nb) - 1ull)) << n;
        n += nb;
        if (n >= 32u) { const uint32_t w = (uint32_t)acc; if (len + 4u <= cap) __builtin_memcpy(p + len, &w, 4); len += 4u; acc >>= 32; n -= 32u; }
    }
    __device__ inline uint32_t flush()
    {
        while (n > 0u) { if (len < cap) p[len] = (uint8_t)acc; len++; acc >>= 8; n = n > 8u ? n - 8u : 0u; }
        acc = 0;
        return len;
    }
    __device__ inline uint32_t close() { add(1, 1); return flush(); }
};

/* ------------------------------------------------------------ FSE (compression side); wave-uniform */
__device__ inline int fse_min_log(uint32_t n, uint32_t max_sym)
{
    const int a = (int)hbit(n) + 1, b = (int)hbit(max_sym) + 2;
    return a < b ? a : b;
}
__device__ inline int fse_optimal_log(int max_log, uint32_t n, uint32_t max_sym, int minus)
{
    /* the library's highbit32(srcSize - 1) - minus is unsigned: below 2^minus + 1 symbols it wraps and limits nothing */
    const int max_src = (n > 1u ? (int)hbit(n - 1u) : 0) - minus;
    int log = max_log;
    const int min_bits = fse_min_log(n, max_sym);
    if (max_src >= 0 && max_src < log) log = max_src;
    if (min_bits > log) log = min_bits;
    if (log < 5) log = 5;
    if (log > 12) log = 12;
    return log;
}

__device__ int fse_norm_m2(int16_t *norm, int log, const uint32_t *count, uint32_t total_in, uint32_t max_sym,
                           int16_t low_prob)
{
    const int16_t NYA = -2;
    uint64_t total = total_in;
    uint32_t distributed = 0;
    const uint32_t low_thr = (uint32_t)(total >> log);
    uint32_t low_one = (uint32_t)((total * 3u) >> (log + 1));
    for (uint32_t s = 0; s <= max_sym; s++) {
        const uint32_t c = count[s];
        if (c == 0u) { norm[s] = 0; continue; }
        if (c <= low_thr) { norm[s] = low_prob; distributed++; total -= c; continue; }
        if (c <= low_one) { norm[s] = 1; distributed++; total -= c; continue; }
        norm[s] = NYA;
    }
    uint32_t to_dist = (1u << log) - distributed;
    if (to_dist == 0u) return 0;
    if ((total / to_dist) > low_one) {
        low_one = (uint32_t)((total * 3u) / (to_dist * 2u));
        for (uint32_t s = 0; s <= max_sym; s++)
            if (norm[s] == NYA && count[s] <= low_one) { norm[s] = 1; distributed++; total -= count[s]; }
        to_dist = (1u << log) - distributed;
    }
    if (distributed == max_sym + 1u) {
        uint32_t mv = 0, mc = 0;
        for (uint32_t s = 0; s <= max_sym; s++) if (count[s] > mc) { mv = s; mc = count[s]; }
        norm[mv] = (int16_t)(norm[mv] + (int16_t)to_dist);
        return 0;
    }
    if (total == 0u) {
        for (uint32_t s = 0; to_dist > 0u; s = (s + 1u) % (max_sym + 1u))
            if (norm[s] > 0) { to_dist--; norm[s] = (int16_t)(norm[s] + 1); }
        return 0;
    }
    {
        const uint64_t vlog = 62u - (uint64_t)log, mid = (1ull << (vlog - 1u)) - 1u;
        const uint64_t rstep = (((1ull << vlog) * to_dist) + mid) / total;
        uint64_t tmp = mid;
        for (uint32_t s = 0; s <= max_sym; s++) {
            if (norm[s] == NYA) {
                const uint64_t end = tmp + (uint64_t)count[s] * rstep;
                const uint32_t w = (uint32_t)(end >> vlog) - (uint32_t)(tmp >> vlog);
                if (w < 1u) return -1;
                norm[s] = (int16_t)w;
                tmp = end;
            }
        }
    }
    return 0;
}

/* FSE_normalizeCount (libzstd 1.4.8: low-probability symbols get -1 only when use_low_prob) */
__device__ int fse_normalize(int16_t *norm, int log, const uint32_t *count, uint32_t total, uint32_t max_sym,
                             bool use_low_prob)
{
    const uint32_t rtb[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
    const int16_t low_prob = use_low_prob ? -1 : 1;
    const uint64_t scale = 62u - (uint64_t)log, step = (1ull << 62) / total, vstep = 1ull << (scale - 20u);
    int still = 1 << log;
    uint32_t largest = 0;
    int16_t largest_p = 0;
    const uint32_t low_thr = total >> log;
    if (log < fse_min_log(total, max_sym)) return -1;
    for (uint32_t s = 0; s <= max_sym; s++) {
        const uint32_t c = count[s];
        if (c == total) return 0;
        if (c == 0u) { norm[s] = 0; continue; }
        if (c <= low_thr) { norm[s] = low_prob; still--; }
        else {
            int16_t proba = (int16_t)(((uint64_t)c * step) >> scale);
            if (proba < 8) {
                const uint64_t rest = vstep * rtb[proba];
                proba = (int16_t)(proba + ((((uint64_t)c * step) - ((uint64_t)proba << scale)) > rest ? 1 : 0));
            }
            if (proba > largest_p) { largest_p = proba; largest = s; }
            norm[s] = proba;
            still -= proba;
        }
    }
    if (-still >= (norm[largest] >> 1)) { if (fse_norm_m2(norm, log, count, total, max_sym, low_prob)) return -1; }
    else norm[largest] = (int16_t)(norm[largest] + (int16_t)still);
    return log;
}

__device__ uint32_t fse_write_ncount(uint8_t *dst, const int16_t *norm, uint32_t max_sym, int log)
{
    BitW b;
    b.init(dst);
    const int table_size = 1 << log;
    int remaining = table_size + 1, threshold = table_size, nb = log + 1;
    bool prev0 = false;
    uint32_t sym = 0;
    const uint32_t alpha = max_sym + 1u;
    b.add((uint64_t)(log - 5), 4);
    while (sym < alpha && remaining > 1) {
        if (prev0) {
            uint32_t start = sym;
            while (sym < alpha && !norm[sym]) sym++;
            if (sym == alpha) break;
            while (sym >= start + 24u) { start += 24u; b.add(0xFFFF, 16); }
            while (sym >= start + 3u) { start += 3u; b.add(3, 2); }
            b.add(sym - start, 2);
        }
        int count = norm[sym++];
        const int max = (2 * threshold - 1) - remaining;
        remaining -= count < 0 ? -count : count;
        count++;
        if (count >= threshold) count += max;
        b.add((uint64_t)count, (uint32_t)(nb - (count < max ? 1 : 0)));
        prev0 = (count == 1);
        if (remaining < 1) return 0;
        while (remaining < threshold) { nb--; threshold >>= 1; }
    }
    if (remaining != 1) return 0;
    return b.flush();
}

__device__ void fse_build_ct(FseCt &ct, const int16_t *norm, uint32_t max_sym, int log, uint32_t *cumul, uint8_t *cell)
{
    const uint32_t size = 1u << log, mask = size - 1u, step = (size >> 1) + (size >> 3) + 3u;
    uint32_t high = size - 1u, pos = 0;
    ct.log = log;
    ct.max_sym = max_sym;
    cumul[0] = 0;
    for (uint32_t u = 1; u <= max_sym + 1u; u++) {
        const int c = norm[u - 1u];
        if (c == -1) { cumul[u] = cumul[u - 1u] + 1u; cell[high--] = (uint8_t)(u - 1u); }
        else cumul[u] = cumul[u - 1u] + (uint32_t)c;
    }
    cumul[max_sym + 1u] = size + 1u;
    for (uint32_t u = 0; u <= max_sym; u++) {
        const int c = norm[u];
        for (int i = 0; i < c; i++) {
            cell[pos] = (uint8_t)u;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    for (uint32_t u = 0; u < size; u++) {
        const uint32_t s = cell[u];
        const uint32_t k = cumul[s];
        cumul[s] = k + 1u;
        ct.state[k] = (uint16_t)(size + u);
    }
    uint32_t total = 0;
    for (uint32_t s = 0; s <= max_sym; s++) {
        const int c = norm[s];
        if (c == 0) { ct.dnb[s] = ((uint32_t)(log + 1) << 16) - (1u << log); ct.dfind[s] = 0; }
        else if (c == -1 || c == 1) { ct.dnb[s] = ((uint32_t)log << 16) - (1u << log); ct.dfind[s] = (int32_t)total - 1; total++; }
        else {
            const uint32_t max_out = (uint32_t)log - hbit((uint32_t)c - 1u);
            ct.dnb[s] = (max_out << 16) - ((uint32_t)c << max_out);
            ct.dfind[s] = (int32_t)total - c;
            total += (uint32_t)c;
        }
    }
}
__device__ inline void fse_build_ct_rle(FseCt &ct, uint32_t sym)
{
    ct.log = 0;
    ct.max_sym = sym;
    ct.state[0] = 0; ct.state[1] = 0;
    ct.dnb[sym] = 0; ct.dfind[sym] = 0;
}
__device__ inline uint32_t fse_init_state(const FseCt &ct, uint32_t sym)
{
    const uint32_t dnb = ct.dnb[sym];
    const uint32_t nb = (dnb + (1u << 15)) >> 16;
    const uint32_t v = (nb << 16) - dnb;
    return ct.state[(int32_t)(v >> nb) + ct.dfind[sym]];
}
__device__ inline uint32_t fse_encode(BitW &b, const FseCt &ct, uint32_t state, uint32_t sym)
{
    const uint32_t nb = (state + ct.dnb[sym]) >> 16;
    b.add(state, nb);
    return ct.state[(int32_t)(state >> nb) + ct.dfind[sym]];
}

/* ------------------------------------------------------------ Huffman (compression side); wave-uniform */
#define HN(i) ((i) + 1) /* library index -> LDS index */

__device__ uint32_t huf_set_max_height(EncLds &L, uint32_t last, uint32_t max_nb)
{
    const uint32_t largest = L.nnb[HN(last)];
    if (largest <= max_nb) return largest;
    int total = 0, n = (int)last;
    const uint32_t base = 1u << (largest - max_nb);
    while (L.nnb[HN(n)] > max_nb) { total += (int)(base - (1u << (largest - L.nnb[HN(n)]))); L.nnb[HN(n)] = (uint8_t)max_nb; n--; }
    while (L.nnb[HN(n)] == max_nb) n--;
    total >>= (largest - max_nb);
    const uint32_t none = 0xF0F0F0F0u;
    uint32_t *rank_last = L.cumul; /* scratch: kHufLogMaxE + 2 entries */
    for (int k = 0; k < kHufLogMaxE + 2; k++) rank_last[k] = none;
    {
        uint32_t cur = max_nb;
        for (int pos = n; pos >= 0; pos--) {
            const uint32_t nb = L.nnb[HN(pos)];
            if (nb >= cur) continue;
            cur = nb;
            rank_last[max_nb - cur] = (uint32_t)pos;
        }
    }
    while (total > 0) {
        uint32_t dec = hbit((uint32_t)total) + 1u;
        for (; dec > 1u; dec--) {
            const uint32_t hp = rank_last[dec], lp = rank_last[dec - 1u];
            if (hp == none) continue;
            if (lp == none) break;
            if (L.ncount[HN(hp)] <= 2u * L.ncount[HN(lp)]) break;
        }
        while (dec <= (uint32_t)kHufLogMaxE && rank_last[dec] == none) dec++;
        total -= 1 << (dec - 1u);
        if (rank_last[dec - 1u] == none) rank_last[dec - 1u] = rank_last[dec];
        L.nnb[HN(rank_last[dec])] = (uint8_t)(L.nnb[HN(rank_last[dec])] + 1);
        if (rank_last[dec] == 0u) rank_last[dec] = none;
        else {
            rank_last[dec] = rank_last[dec] - 1u;
            if (L.nnb[HN(rank_last[dec])] != max_nb - dec) rank_last[dec] = none;
        }
    }
    while (total < 0) {
        if (rank_last[1] == none) {
            while (L.nnb[HN(n)] == max_nb) n--;
            L.nnb[HN(n + 1)] = (uint8_t)(L.nnb[HN(n + 1)] - 1);
            rank_last[1] = (uint32_t)(n + 1);
            total++;
            continue;
        }
        L.nnb[HN(rank_last[1] + 1u)] = (uint8_t)(L.nnb[HN(rank_last[1] + 1u)] - 1);
        rank_last[1] = rank_last[1] + 1u;
        total++;
    }
    return max_nb;
}

/* HUF_buildCTable from L.hist[0..max_sym] into L.hval / L.hnb; returns the longest code length */
__device__ uint32_t huf_build(EncLds &L, uint32_t max_sym, uint32_t max_nb, uint32_t lane)
{
    for (uint32_t i = lane; i < 516u; i += 64u) { L.ncount[i] = 0; L.nparent[i] = 0; L.nbyte[i] = 0; L.nnb[i] = 0; }
    __builtin_amdgcn_wave_barrier();
    /* sort by decreasing count, ties by increasing symbol (bucketed insertion sort) */
    {
        uint32_t *base = L.cumul;        /* 33 entries */
        uint32_t *cur = L.cumul + 33;    /* 33 entries */
        for (int k = 0; k < 33; k++) base[k] = 0;
        for (uint32_t n = 0; n <= max_sym; n++) { const uint32_t r = hbit(L.hist[n] + 1u); base[r] = base[r] + 1u; }
        for (int k = 30; k > 0; k--) base[k - 1] = base[k - 1] + base[k];
        for (int k = 0; k < 32; k++) cur[k] = base[k];
        for (uint32_t n = 0; n <= max_sym; n++) {
            const uint32_t c = L.hist[n], r = hbit(c + 1u) + 1u;
            uint32_t pos = cur[r];
            cur[r] = pos + 1u;
            while (pos > base[r] && c > L.ncount[HN(pos - 1u)]) {
                L.ncount[HN(pos)] = L.ncount[HN(pos - 1u)];
                L.nbyte[HN(pos)] = L.nbyte[HN(pos - 1u)];
                pos--;
            }
            L.ncount[HN(pos)] = c;
            L.nbyte[HN(pos)] = (uint8_t)n;
        }
    }
    int non_null = (int)max_sym;
    while (L.ncount[HN(non_null)] == 0u) non_null--;
    int low_s = non_null, node_nb = 256, low_n = 256;
    const int node_root = node_nb + low_s - 1;
    L.ncount[HN(node_nb)] = L.ncount[HN(low_s)] + L.ncount[HN(low_s - 1)];
    L.nparent[HN(low_s)] = (uint16_t)node_nb;
    L.nparent[HN(low_s - 1)] = (uint16_t)node_nb;
    node_nb++; low_s -= 2;
    for (int n = node_nb; n <= node_root; n++) L.ncount[HN(n)] = 1u << 30;
    L.ncount[0] = 1u << 31; /* fake entry, strong barrier (library index -1) */
    while (node_nb <= node_root) {
        const int n1 = (L.ncount[HN(low_s)] < L.ncount[HN(low_n)]) ? low_s-- : low_n++;
        const int n2 = (L.ncount[HN(low_s)] < L.ncount[HN(low_n)]) ? low_s-- : low_n++;
        L.ncount[HN(node_nb)] = L.ncount[HN(n1)] + L.ncount[HN(n2)];
        L.nparent[HN(n1)] = (uint16_t)node_nb;
        L.nparent[HN(n2)] = (uint16_t)node_nb;
        node_nb++;
    }
    L.nnb[HN(node_root)] = 0;
    for (int n = node_root - 1; n >= 256; n--) L.nnb[HN(n)] = (uint8_t)(L.nnb[HN(L.nparent[HN(n)])] + 1);
    for (int n = 0; n <= non_null; n++) L.nnb[HN(n)] = (uint8_t)(L.nnb[HN(L.nparent[HN(n)])] + 1);
    max_nb = huf_set_max_height(L, (uint32_t)non_null, max_nb);
    {
        uint32_t *per_rank = L.cumul;       /* 14 entries */
        uint32_t *val_rank = L.cumul + 16;  /* 14 entries */
        for (int k = 0; k < 14; k++) { per_rank[k] = 0; val_rank[k] = 0; }
        for (int n = 0; n <= non_null; n++) { const uint32_t nb = L.nnb[HN(n)]; per_rank[nb] = per_rank[nb] + 1u; }
        {
            uint32_t min = 0;
            for (int n = (int)max_nb; n > 0; n--) { val_rank[n] = min; min = (min + per_rank[n]) & 0xFFFFu; min >>= 1; }
        }
        for (uint32_t i = lane; i < 256u; i += 64u) { L.hval[i] = 0; L.hnb[i] = 0; }
        __builtin_amdgcn_wave_barrier();
        const int alpha = (int)max_sym + 1;
        for (int n = 0; n < alpha; n++) L.hnb[L.nbyte[HN(n)]] = L.nnb[HN(n)];
        for (int n = 0; n < alpha; n++) { const uint32_t nb = L.hnb[n]; const uint32_t v = val_rank[nb]; val_rank[nb] = v + 1u; L.hval[n] = (uint16_t)v; }
    }
    return max_nb;
}

/* HUF_compressWeights: 0 = not compressible, 1 = RLE, else size */
__device__ uint32_t huf_compress_weights(EncLds &L, uint8_t *dst, uint32_t n)
{
    uint32_t *count = L.cumul + 40; /* 13 entries */
    uint32_t max_sym = (uint32_t)kHufLogMaxE, max_count = 0;
    if (n <= 1u) return 0;
    for (uint32_t s = 0; s <= (uint32_t)kHufLogMaxE; s++) count[s] = 0;
    for (uint32_t i = 0; i < n; i++) { const uint32_t wv = L.wts[i]; count[wv] = count[wv] + 1u; }
    while (!count[max_sym]) max_sym--;
    for (uint32_t s = 0; s <= max_sym; s++) if (count[s] > max_count) max_count = count[s];
    if (max_count == n) return 1;
    if (max_count == 1u) return 0;
    const int log = fse_optimal_log(6, n, max_sym, 2);
    if (fse_normalize(L.norm, log, count, n, max_sym, false) <= 0) return 0;
    const uint32_t hsz = fse_write_ncount(dst, L.norm, max_sym, log);
    if (!hsz) return 0;
    fse_build_ct(L.ll, L.norm, max_sym, log, L.cumul, L.cell);
    if (n <= 2u) return 0;
    BitW b;
    b.init(dst + hsz);
    uint32_t ip = n, s1, s2;
    if (n & 1u) {
        s1 = fse_init_state(L.ll, L.wts[--ip]);
        s2 = fse_init_state(L.ll, L.wts[--ip]);
        s1 = fse_encode(b, L.ll, s1, L.wts[--ip]);
    } else {
        s2 = fse_init_state(L.ll, L.wts[--ip]);
        s1 = fse_init_state(L.ll, L.wts[--ip]);
    }
    while (ip > 0u) {
        s2 = fse_encode(b, L.ll, s2, L.wts[--ip]);
        s1 = fse_encode(b, L.ll, s1, L.wts[--ip]);
    }
    b.add(s2, (uint32_t)L.ll.log);
    b.add(s1, (uint32_t)L.ll.log);
    return hsz + b.close();
}

/* HUF_writeCTable for L.hval/L.hnb; 0 = cannot be described (literals stay raw) */
__device__ uint32_t huf_write_table(EncLds &L, uint8_t *dst, uint32_t max_sym, uint32_t log, uint32_t lane)
{
    for (uint32_t n = lane; n < max_sym; n += 64u) { const uint32_t nb = L.hnb[n]; L.wts[n] = (uint8_t)(nb ? log + 1u - nb : 0u); }
    __builtin_amdgcn_wave_barrier();
    const uint32_t hsz = huf_compress_weights(L, dst + 1, max_sym);
    if (hsz > 1u && hsz < max_sym / 2u) { if (lane == 0) dst[0] = (uint8_t)hsz; return hsz + 1u; }
    if (max_sym > 128u) return 0;
    if (lane == 0) {
        dst[0] = (uint8_t)(128u + (max_sym - 1u));
        L.wts[max_sym] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t n = lane * 2u; n < max_sym; n += 128u) dst[n / 2u + 1u] = (uint8_t)((L.wts[n] << 4) + L.wts[n + 1u]);
    return (max_sym + 1u) / 2u + 1u;
}

/* ---- Huffman literal streams, all 64 lanes per stream ----
 * A stream is the concatenation of its symbols' codes, last symbol first.  The lanes split it into 64
 * runs of consecutive symbols, count their bits (pass 1), scan the counts into bit offsets, and then each
 * lane packs its run at its offset (pass 2): whole dwords it owns are stored, the first and last (shared
 * with the neighbours, the jump table or the previous stream) are OR-ed in atomically into the zeroed
 * output.  Symbols are fetched 16 at a time. */
__device__ inline uint32_t huf_run_bits(const uint8_t *src, uint32_t top, uint32_t cnt, const uint8_t *nbt)
{
    uint32_t bits = 0, i = 0;
    for (; i + 16u <= cnt; i += 16u) {
        uint4 v;
        __builtin_memcpy(&v, src + top - i - 16u, 16);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; k++) bits += nbt[(w[k >> 2] >> (8 * (k & 3))) & 255u];
    }
    for (; i < cnt; i++) bits += nbt[src[top - 1u - i]];
    return bits;
}

struct OrW { /* bit packer over dwords: first and last dword OR-ed atomically, the ones between stored */
    uint32_t *a;
    uint64_t acc;
    uint32_t nacc;
    bool first;
    __device__ inline void init(uint32_t *base, uint32_t bitpos) { a = base + (bitpos >> 5); acc = 0; nacc = bitpos & 31u; first = true; }
    __device__ inline void add(uint32_t v, uint32_t nb)
    {
        acc |= (uint64_t)v << nacc;
        nacc += nb;
        if (nacc >= 32u) {
            if (first) atomicOr(a, (uint32_t)acc); else *a = (uint32_t)acc;
            first = false;
            a++;
            acc >>= 32;
            nacc -= 32u;
        }
    }
    __device__ inline void finish() { if (nacc) atomicOr(a, (uint32_t)acc); }
};

__device__ inline void huf_run_emit(OrW &o, const uint8_t *src, uint32_t top, uint32_t cnt, const uint16_t *val, const uint8_t *nbt)
{
    uint32_t i = 0;
    for (; i + 16u <= cnt; i += 16u) {
        uint4 v;
        __builtin_memcpy(&v, src + top - i - 16u, 16);
        const uint32_t w[4] = {v.w, v.z, v.y, v.x}; /* highest address first */
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t c = (w[k >> 2] >> (8 * (3 - (k & 3)))) & 255u;
            o.add(val[c], nbt[c]);
        }
    }
    for (; i < cnt; i++) { const uint32_t c = src[top - 1u - i]; o.add(val[c], nbt[c]); }
}

/* HUF_compress1X / 4X body: the n symbols at src with code table (val, nbt) -> dst + hsz.
 * Returns the total size written at dst+hsz (jump table included) + hsz, 0 = not compressible */
__device__ uint32_t huf_encode_streams(EncLds &L, uint8_t *dst, uint32_t hsz, const uint8_t *src, uint32_t n,
                                       const uint16_t *val, const uint8_t *nbt, bool single, uint32_t lane)
{
    const uint32_t nstreams = single ? 1u : 4u;
    const uint32_t seg = single ? n : (n + 3u) / 4u;
    if (!single && n < 12u) return 0;
    /* pass 1: every lane's run in every stream; bit offsets by scan */
    uint32_t off[4] = {0, 0, 0, 0}, bytes[4] = {0, 0, 0, 0}, tot[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t s4 = 0; s4 < 4u; s4++) {
        if (s4 < nstreams) {
            const uint32_t beg = s4 * seg, end = (s4 + 1u == nstreams) ? n : beg + seg;
            const uint32_t m = end - beg, q = (m + 63u) >> 6;
            const uint32_t r0 = lane * q < m ? lane * q : m, r1 = r0 + q < m ? r0 + q : m;
            const uint32_t bits = huf_run_bits(src, end - r0, r1 - r0, nbt);
            const uint32_t inc = scan64_incl(bits);
            off[s4] = inc - bits;
            tot[s4] = lane_get(inc, 63u);
            bytes[s4] = (tot[s4] + 1u + 7u) >> 3; /* + end mark */
        }
    }
    uint32_t op = hsz + (single ? 0u : 6u);
    const uint32_t total = bytes[0] + bytes[1] + bytes[2] + bytes[3];
    if (op + total >= n - 1u) return 0; /* not compressible (same verdict as after writing, without the writes) */
    /* zero the output range (the bytes up to the next dword boundary one by one: the dword before holds
     * the header) */
    {
        uint8_t *z0 = dst + hsz;
        const uint32_t len = (single ? 0u : 6u) + total;
        const uint32_t head = (4u - (uint32_t)(reinterpret_cast<uintptr_t>(z0) & 3u)) & 3u;
        if (lane < head && lane < len) z0[lane] = 0;
        if (len > head) {
            uint32_t *zw = reinterpret_cast<uint32_t *>(z0 + head);
            const uint32_t nw = (len - head + 3u) >> 2;
            for (uint32_t i = lane; i < nw; i += 64u) zw[i] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_s_waitcnt(0);
    }
    if (!single && lane == 0) {
        dst[hsz + 0] = (uint8_t)bytes[0]; dst[hsz + 1] = (uint8_t)(bytes[0] >> 8);
        dst[hsz + 2] = (uint8_t)bytes[1]; dst[hsz + 3] = (uint8_t)(bytes[1] >> 8);
        dst[hsz + 4] = (uint8_t)bytes[2]; dst[hsz + 5] = (uint8_t)(bytes[2] >> 8);
    }
    /* pass 2 */
    uint32_t sofs = op;
#pragma unroll
    for (uint32_t s4 = 0; s4 < 4u; s4++) {
        if (s4 < nstreams) {
            const uint32_t beg = s4 * seg, end = (s4 + 1u == nstreams) ? n : beg + seg;
            const uint32_t m = end - beg, q = (m + 63u) >> 6;
            const uint32_t r0 = lane * q < m ? lane * q : m, r1 = r0 + q < m ? r0 + q : m;
            uint8_t *sp = dst + sofs;
            const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(sp) & 3u);
            uint32_t *base = reinterpret_cast<uint32_t *>(sp - mis);
            OrW o;
            o.init(base, 8u * mis + off[s4]);
            huf_run_emit(o, src, end - r0, r1 - r0, val, nbt);
            o.finish();
            if (lane == 0) { /* end mark */
                const uint32_t bp = 8u * mis + tot[s4];
                atomicOr(base + (bp >> 5), 1u << (bp & 31u));
            }
            sofs += bytes[s4];
        }
    }
    op += total;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (op >= n - 1u) return 0;
    return op;
}

struct HufState { bool prev_valid; bool next_new; unsigned long long *prof; unsigned long long t; uint32_t strat; };
/* diagnostic phase stamps (CRYO_ZSTD_STATS): bucket k gets the cycles since the previous stamp */
__device__ inline void zprof(HufState &hs, int k)
{
    if (hs.prof) { const unsigned long long now = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) atomicAdd(&hs.prof[k], now - hs.t); hs.t = now; }
}

/* n bytes from one place in memory to another by the wave: 16 bytes per lane, four such loads in flight (a byte per lane and
 * trip, 2 048 trips for an incompressible block's literals, raw block and last literals, was most of what `random` cost) */
__device__ inline void wave_copy_bytes(uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t lane)
{
    uint32_t o = 0;
    for (; o + 4096u <= n; o += 4096u) {
        uint4 a, b, c, d;
        __builtin_memcpy(&a, src + o + 16u * lane, 16);
        __builtin_memcpy(&b, src + o + 1024u + 16u * lane, 16);
        __builtin_memcpy(&c, src + o + 2048u + 16u * lane, 16);
        __builtin_memcpy(&d, src + o + 3072u + 16u * lane, 16);
        __builtin_memcpy(dst + o + 16u * lane, &a, 16);
        __builtin_memcpy(dst + o + 1024u + 16u * lane, &b, 16);
        __builtin_memcpy(dst + o + 2048u + 16u * lane, &c, 16);
        __builtin_memcpy(dst + o + 3072u + 16u * lane, &d, 16);
    }
    for (; o + 1024u <= n; o += 1024u) {
        uint4 a;
        __builtin_memcpy(&a, src + o + 16u * lane, 16);
        __builtin_memcpy(dst + o + 16u * lane, &a, 16);
    }
    for (uint32_t i = o + lane; i < n; i += 64u) dst[i] = src[i];
}

/* ZSTD_compressLiterals; returns the literals-section size.  Updates hs.next_new. */
__device__ uint32_t compress_literals(EncLds &L, uint8_t *dst, const uint8_t *src, uint32_t n, HufState &hs,
                                      bool disable, uint32_t lane)
{
    hs.next_new = false;
    const uint32_t fl = 1u + (n > 31u) + (n > 4095u);
    auto raw = [&]() {
        if (lane == 0) {
            if (fl == 1u) dst[0] = (uint8_t)(0u + (n << 3));
            else if (fl == 2u) { const uint32_t h = 0u + (1u << 2) + (n << 4); dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); }
            else { const uint32_t h = 0u + (3u << 2) + (n << 4); dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16); }
        }
        wave_copy_bytes(dst + fl, src, n, lane);
        return fl + n;
    };
    if (disable || n <= 63u) return raw();
    const uint32_t lh = 3u + (n >= 1024u) + (n >= 16384u);
    const bool single = n < 256u;
    /* histogram */
    for (uint32_t i = lane; i < 256u; i += 64u) L.hist[i] = 0;
    __builtin_amdgcn_wave_barrier();
    {   /* 16 bytes per lane and load, four loads in flight (a byte per lane and trip: 2 048 trips for 128 KiB of literals) */
        auto add16 = [&](const uint4 v) {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                atomicAdd(&L.hist[w[k] & 255u], 1u);
                atomicAdd(&L.hist[(w[k] >> 8) & 255u], 1u);
                atomicAdd(&L.hist[(w[k] >> 16) & 255u], 1u);
                atomicAdd(&L.hist[w[k] >> 24], 1u);
            }
        };
        uint32_t o = 0;
        for (; o + 4096u <= n; o += 4096u) {
            uint4 a, b, c, d;
            __builtin_memcpy(&a, src + o + 16u * lane, 16);
            __builtin_memcpy(&b, src + o + 1024u + 16u * lane, 16);
            __builtin_memcpy(&c, src + o + 2048u + 16u * lane, 16);
            __builtin_memcpy(&d, src + o + 3072u + 16u * lane, 16);
            add16(a); add16(b); add16(c); add16(d);
        }
        for (; o + 1024u <= n; o += 1024u) {
            uint4 a;
            __builtin_memcpy(&a, src + o + 16u * lane, 16);
            add16(a);
        }
        for (uint32_t i = o + lane; i < n; i += 64u) atomicAdd(&L.hist[src[i]], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t max_sym = 255, largest = 0;
    while (!L.hist[max_sym]) max_sym--;
    for (uint32_t s = 0; s <= max_sym; s++) { const uint32_t c = L.hist[s]; if (c > largest) largest = c; }
    max_sym = uni(max_sym);
    largest = uni(largest);
    uint32_t c = 0;
    bool reused = false;
    if (largest == n) c = 1; /* RLE */
    else if (largest <= (n >> 7) + 4u) c = 0;
    else {
        bool mode = hs.prev_valid;
        if (mode) { /* HUF_validateCTable */
            bool bad = false;
            for (uint32_t s = 0; s <= max_sym; s++) bad |= (L.hist[s] != 0u) && (L.pnb[s] == 0u);
            if (bad) mode = false;
        }
        if (hs.strat < 4u && n <= 1024u && mode) { reused = true; /* preferRepeat: below `lazy` only */ c = huf_encode_streams(L, dst + lh, 0, src, n, L.pval, L.pnb, single, lane); }
        else {
            uint32_t log = (uint32_t)fse_optimal_log(11, n, max_sym, 1);
            zprof(hs, 3);
            log = huf_build(L, max_sym, log, lane);
            __builtin_amdgcn_wave_barrier();
            const uint32_t hsz = huf_write_table(L, dst + lh, max_sym, log, lane);
            zprof(hs, 4);
            if (hsz == 0u) c = 0;
            else {
                bool use_old = false;
                if (mode) {
                    uint32_t old_bits = 0, new_bits = 0;
                    for (uint32_t s = 0; s <= max_sym; s++) { const uint32_t k = L.hist[s]; old_bits += L.pnb[s] * k; new_bits += L.hnb[s] * k; }
                    use_old = ((old_bits >> 3) <= hsz + (new_bits >> 3)) || (hsz + 12u >= n);
                }
                if (use_old) { reused = true; c = huf_encode_streams(L, dst + lh, 0, src, n, L.pval, L.pnb, single, lane); }
                else if (hsz + 12u >= n) c = 0;
                else { hs.next_new = true; c = huf_encode_streams(L, dst + lh, hsz, src, n, L.hval, L.hnb, single, lane); }
            }
        }
    }
    zprof(hs, 5);
    const uint32_t gain = (n >> (hs.strat >= 8u ? hs.strat - 1u : 6u)) + 2u; /* ZSTD_minGain: btultra, btultra2 accept smaller gains */
    if (c == 0u || c >= n - gain) { hs.next_new = false; return raw(); }
    if (c == 1u) {
        hs.next_new = false;
        if (lane == 0) {
            if (fl == 1u) dst[0] = (uint8_t)(1u + (n << 3));
            else if (fl == 2u) { const uint32_t h = 1u + (1u << 2) + (n << 4); dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); }
            else { const uint32_t h = 1u + (3u << 2) + (n << 4); dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16); }
            dst[fl] = src[0];
        }
        return fl + 1u;
    }
    if (lane == 0) {
        const uint32_t ht = reused ? 3u : 2u;
        if (lh == 3u) { const uint32_t h = ht + ((single ? 0u : 1u) << 2) + (n << 4) + (c << 14); dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16); }
        else if (lh == 4u) { const uint32_t h = ht + (2u << 2) + (n << 4) + (c << 18); dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16); dst[3] = (uint8_t)(h >> 24); }
        else { const uint32_t h = ht + (3u << 2) + (n << 4) + (c << 22); dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16); dst[3] = (uint8_t)(h >> 24); dst[4] = (uint8_t)(c >> 10); }
    }
    return lh + c;
}

/* ------------------------------------------------------------ sequences */
__constant__ uint8_t kELLBits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6,
    7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ uint8_t kEMLBits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ int16_t kELLDef[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3,
    2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ int16_t kEMLDef[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__constant__ int16_t kEOFDef[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1,
    -1, -1, -1};
__constant__ uint8_t kLLCode[64] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 16, 17, 17, 18, 18, 19,
    19, 20, 20, 20, 20, 21, 21, 21, 21, 22, 22, 22, 22, 22, 22, 22, 22, 23, 23, 23, 23, 23, 23, 23, 23, 24, 24,
    24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24};
__constant__ uint8_t kMLCode[128] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22,
    23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 32, 33, 33, 34, 34, 35, 35, 36, 36, 36, 36, 37, 37, 37, 37, 38, 38,
    38, 38, 38, 38, 38, 38, 39, 39, 39, 39, 39, 39, 39, 39, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40,
    40, 40, 40, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 42, 42, 42, 42, 42, 42, 42, 42,
    42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42};

enum { SET_BASIC = 0, SET_RLE = 1, SET_COMPRESSED = 2, SET_REPEAT = 3 };

/* kInverseProbabilityLog256[x] = (unsigned)(-log2(x / 256.) * 256) of the library's cost model */
__constant__ uint16_t kInvProbLog256[256] = {0, 2048, 1792, 1642, 1536, 1453, 1386, 1329, 1280, 1236, 1197, 1162, 1130, 1100, 1073, 1047,
    1024, 1001, 980, 960, 941, 923, 906, 889, 874, 859, 844, 830, 817, 804, 791, 779, 768, 756, 745, 734, 724, 714, 704, 694, 685, 676,
    667, 658, 650, 642, 633, 626, 618, 610, 603, 595, 588, 581, 574, 567, 561, 554, 548, 542, 535, 529, 523, 517, 512, 506, 500, 495, 489,
    484, 478, 473, 468, 463, 458, 453, 448, 443, 438, 434, 429, 424, 420, 415, 411, 407, 402, 398, 394, 390, 386, 382, 377, 373, 370, 366,
    362, 358, 354, 350, 347, 343, 339, 336, 332, 329, 325, 322, 318, 315, 311, 308, 305, 302, 298, 295, 292, 289, 286, 282, 279, 276, 273,
    270, 267, 264, 261, 258, 256, 253, 250, 247, 244, 241, 239, 236, 233, 230, 228, 225, 222, 220, 217, 215, 212, 209, 207, 204, 202, 199,
    197, 194, 192, 190, 187, 185, 182, 180, 178, 175, 173, 171, 168, 166, 164, 162, 159, 157, 155, 153, 151, 149, 146, 144, 142, 140, 138,
    136, 134, 132, 130, 128, 126, 123, 121, 119, 117, 115, 114, 112, 110, 108, 106, 104, 102, 100, 98, 96, 94, 93, 91, 89, 87, 85, 83, 82,
    80, 78, 76, 74, 73, 71, 69, 67, 66, 64, 62, 61, 59, 57, 55, 54, 52, 50, 49, 47, 46, 44, 42, 41, 39, 37, 36, 34, 33, 31, 30, 28, 26, 25,
    23, 22, 20, 19, 17, 16, 14, 13, 11, 10, 8, 7, 5, 4, 2, 1};

constexpr uint64_t kCostErr = ~0ull;

/* ZSTD_entropyCost / ZSTD_crossEntropyCost / ZSTD_fseBitCost / ZSTD_NCountCost (libzstd 1.4.8 zstd_compress_sequences.c;
 * oracle/zstd_enc_oracle.c entropy_cost .. ncount_cost).  Wave-uniform like the rest of the table stage. */
__device__ uint64_t entropy_cost(const uint32_t *count, uint32_t max, uint32_t total)
{
    uint32_t cost = 0;
    for (uint32_t s = 0; s <= max; s++) {
        const uint32_t c = count[s];
        uint32_t norm = (256u * c) / total;
        if (c != 0u && norm == 0u) norm = 1u;
        cost += c * kInvProbLog256[norm];
    }
    return cost >> 8;
}
__device__ uint64_t cross_entropy_cost(const int16_t *norm, uint32_t acc_log, const uint32_t *count, uint32_t max)
{
    const uint32_t shift = 8u - acc_log;
    uint64_t cost = 0;
    for (uint32_t s = 0; s <= max; s++) {
        const uint32_t nacc = norm[s] != -1 ? (uint32_t)norm[s] : 1u;
        cost += (uint64_t)count[s] * kInvProbLog256[nacc << shift];
    }
    return cost >> 8;
}
__device__ uint64_t fse_bit_cost(const FseCt *ct, const uint32_t *count, uint32_t max)
{
    const uint32_t tlog = (uint32_t)uni((uint32_t)ct->log), bad = (tlog + 1u) << 8;
    uint64_t cost = 0;
    if (uni(ct->max_sym) < max) return kCostErr;
    for (uint32_t s = 0; s <= max; s++) {
        const uint32_t dnb = uni(ct->dnb[s]);
        const uint32_t min_nb = dnb >> 16;
        const uint32_t threshold = (min_nb + 1u) << 16;
        const uint32_t from_thr = threshold - (dnb + (1u << tlog));
        const uint32_t norm_from_thr = (from_thr << 8) >> tlog;
        const uint32_t bit_cost = (min_nb + 1u) * 256u - norm_from_thr;
        if (count[s] == 0u) continue;
        if (bit_cost >= bad) return kCostErr;
        cost += (uint64_t)count[s] * bit_cost;
    }
    return cost >> 8;
}
/* `scratch`: where the description would go (at least 512 bytes of the output bound are still free there) */
__device__ uint64_t ncount_cost(int16_t *norm, uint8_t *scratch, const uint32_t *count, uint32_t max, uint32_t nseq, int fse_log)
{
    const int log = fse_optimal_log(fse_log, nseq, max, 2);
    if (fse_normalize(norm, log, count, nseq, max, nseq >= 2048u) <= 0) return kCostErr;
    const uint32_t sz = fse_write_ncount(scratch, norm, max, log);
    return sz ? sz : kCostErr;
}

/* ZSTD_selectEncodingType.  Strategies below `lazy`: thresholds; `lazy` and above: estimated costs, and the previous
 * block's table may be repeated.  *rep_mode: 0 none, 1 check (the previous compressed block left a usable table). */
__device__ int select_type(int16_t *norm, uint8_t *scratch, const uint32_t *count, uint32_t max, uint32_t most, uint32_t nseq, int fse_log,
                           const FseCt *prev, int *rep_mode, const int16_t *def_norm, int def_log, bool def_allowed, uint32_t strat)
{
    if (most == nseq) { *rep_mode = 0; return (def_allowed && nseq <= 2u) ? SET_BASIC : SET_RLE; }
    if (strat < 4u) {
        if (def_allowed) {
            const uint32_t dyn_min = ((1u << def_log) * (10u - strat)) >> 3; /* ZSTD_fast = 1, ZSTD_dfast = 2, ZSTD_greedy = 3 */
            if (nseq < dyn_min || most < (nseq >> (def_log - 1))) { *rep_mode = 0; return SET_BASIC; }
        }
    } else {
        const uint64_t basic = def_allowed ? cross_entropy_cost(def_norm, (uint32_t)def_log, count, max) : kCostErr;
        const uint64_t repeat = *rep_mode != 0 ? fse_bit_cost(prev, count, max) : kCostErr;
        const uint64_t nc = ncount_cost(norm, scratch, count, max, nseq, fse_log);
        const uint64_t compressed = (nc << 3) + entropy_cost(count, max, nseq);
        if (basic <= repeat && basic <= compressed) { *rep_mode = 0; return SET_BASIC; }
        if (repeat <= compressed) return SET_REPEAT;
    }
    *rep_mode = 1;
    return SET_COMPRESSED;
}

/* histogram of `codes` into L.hist (lane-parallel); returns max symbol and most frequent count */
__device__ void hist_codes(EncLds &L, const uint8_t *codes, uint32_t nseq, uint32_t max_in, uint32_t *max_out,
                           uint32_t *most_out, uint32_t lane)
{
    for (uint32_t i = lane; i < 64u; i += 64u) L.hist[i] = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < nseq; i += 64u) atomicAdd(&L.hist[codes[i]], 1u);
    __builtin_amdgcn_wave_barrier();
    uint32_t max = max_in, most = 0;
    while (!L.hist[max]) max--;
    for (uint32_t s = 0; s <= max; s++) { const uint32_t c = L.hist[s]; if (c > most) most = c; }
    *max_out = uni(max);
    *most_out = uni(most);
}

/* ZSTD_buildCTable; returns bytes of table description, 0xFFFFFFFF on error */
__device__ uint32_t build_ctable(EncLds &L, uint8_t *dst, FseCt &ct, int fse_log, int type, uint32_t max,
                                 const uint8_t *codes, uint32_t nseq, const int16_t *def_norm, int def_log,
                                 uint32_t def_max, const FseCt *prev, uint32_t lane)
{
    if (type == SET_REPEAT) { /* the previous block's table, kept in the workgroup's workspace */
        const uint32_t *from = reinterpret_cast<const uint32_t *>(prev);
        uint32_t *to = reinterpret_cast<uint32_t *>(&ct);
        for (uint32_t i = lane; i < sizeof(FseCt) / 4u; i += 64u) to[i] = from[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return 0;
    }
    if (type == SET_RLE) { fse_build_ct_rle(ct, max); if (lane == 0) dst[0] = codes[0]; return 1; }
    if (type == SET_BASIC) {
        for (uint32_t i = 0; i <= def_max; i++) L.norm[i] = def_norm[i];
        fse_build_ct(ct, L.norm, def_max, def_log, L.cumul, L.cell);
        return 0;
    }
    uint32_t n1 = nseq;
    const int log = fse_optimal_log(fse_log, nseq, max, 2);
    const uint32_t lastc = uni(codes[nseq - 1u]);
    if (L.hist[lastc] > 1u) { L.hist[lastc] = L.hist[lastc] - 1u; n1--; }
    if (fse_normalize(L.norm, log, L.hist, n1, max, n1 >= 2048u) <= 0) return 0xFFFFFFFFu;
    const uint32_t sz = fse_write_ncount(dst, L.norm, max, log);
    if (!sz) return 0xFFFFFFFFu;
    fse_build_ct(ct, L.norm, max, log, L.cumul, L.cell);
    return sz;
}

/* sequence tables + repeat modes that a compressed block leaves to the next one of its frame (`lazy` and above):
 * prev[0..2] = LL, OF, ML tables in the workgroup's workspace; rep = modes confirmed by the last compressed block,
 * nrep = the ones this block would leave */
struct SeqTabs { FseCt *prev; int rep[3]; int nrep[3]; bool built; /* this block built (or repeated) its three tables */ };

/* literals + sequences -> compressed block body at dst; 0 = emit a raw block */
/* seqs / lits: what the match finder left in the workgroup's workspace; codes: 3 x kMaxSeq bytes of scratch for the
 * LL / OF / ML codes */
__device__ uint32_t compress_sequences(EncLds &L, uint8_t *dst, const uint2 *seqs, const uint8_t *lits, uint8_t *codes, uint32_t nseq, uint32_t nlit,
                                       uint32_t src_size, uint32_t long_pos, uint32_t long_kind, HufState &hs, SeqTabs &tb,
                                       bool disable_lit, uint32_t lane)
{
    tb.nrep[0] = tb.rep[0]; tb.nrep[1] = tb.rep[1]; tb.nrep[2] = tb.rep[2];
    tb.built = false;
    uint8_t *llc = codes, *ofc = codes + kMaxSeq, *mlc = codes + 2u * kMaxSeq;
    uint32_t op = compress_literals(L, dst, lits, nlit, hs, disable_lit, lane);
    if (lane == 0) {
        if (nseq < 128u) dst[op] = (uint8_t)nseq;
        else if (nseq < 0x7F00u) { dst[op] = (uint8_t)((nseq >> 8) + 0x80u); dst[op + 1] = (uint8_t)nseq; }
        else { dst[op] = 0xFF; dst[op + 1] = (uint8_t)(nseq - 0x7F00u); dst[op + 2] = (uint8_t)((nseq - 0x7F00u) >> 8); }
    }
    op += nseq < 128u ? 1u : (nseq < 0x7F00u ? 2u : 3u);
    const uint32_t gain = (src_size >> (hs.strat >= 8u ? hs.strat - 1u : 6u)) + 2u;
    if (nseq == 0u) return (op >= src_size - gain) ? 0u : op;
    const uint32_t seq_head = op++;
    for (uint32_t i = lane; i < nseq; i += 64u) {
        const uint2 q = seqs[i];
        const uint32_t ll = q.y & 0xFFFFu, mlb = q.y >> 16;
        llc[i] = (uint8_t)(ll > 63u ? hbit(ll) + 19u : kLLCode[ll]);
        ofc[i] = (uint8_t)hbit(q.x);
        mlc[i] = (uint8_t)(mlb > 127u ? hbit(mlb) + 36u : kMLCode[mlb]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane == 0) {
        if (long_kind == 1u) llc[long_pos] = (uint8_t)kMaxLL;
        if (long_kind == 2u) mlc[long_pos] = (uint8_t)kMaxML;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    uint32_t max, most, last_ncount = 0xFFFFFFFFu;
    hist_codes(L, llc, nseq, kMaxLL, &max, &most, lane);
    const int tll = select_type(L.norm, dst + op, L.hist, max, most, nseq, 9, tb.prev + 0, &tb.nrep[0], kELLDef, 6, true, hs.strat);
    uint32_t sz = build_ctable(L, dst + op, L.ll, 9, tll, max, llc, nseq, kELLDef, 6, kMaxLL, tb.prev + 0, lane);
    if (sz == 0xFFFFFFFFu) return 0;
    if (tll == SET_COMPRESSED) last_ncount = op;
    op += sz;
    hist_codes(L, ofc, nseq, kMaxOff, &max, &most, lane);
    const int tof = select_type(L.norm, dst + op, L.hist, max, most, nseq, 8, tb.prev + 1, &tb.nrep[1], kEOFDef, 5, max <= kDefMaxOff, hs.strat);
    sz = build_ctable(L, dst + op, L.of, 8, tof, max, ofc, nseq, kEOFDef, 5, kDefMaxOff, tb.prev + 1, lane);
    if (sz == 0xFFFFFFFFu) return 0;
    if (tof == SET_COMPRESSED) last_ncount = op;
    op += sz;
    hist_codes(L, mlc, nseq, kMaxML, &max, &most, lane);
    const int tml = select_type(L.norm, dst + op, L.hist, max, most, nseq, 9, tb.prev + 2, &tb.nrep[2], kEMLDef, 6, true, hs.strat);
    sz = build_ctable(L, dst + op, L.ml, 9, tml, max, mlc, nseq, kEMLDef, 6, kMaxML, tb.prev + 2, lane);
    if (sz == 0xFFFFFFFFu) return 0;
    if (tml == SET_COMPRESSED) last_ncount = op;
    op += sz;
    tb.built = true;
    if (lane == 0) dst[seq_head] = (uint8_t)((tll << 6) + (tof << 4) + (tml << 2));
    __builtin_amdgcn_wave_barrier();
    zprof(hs, 6);
    /* Interleaved bitstream, last sequence first.  Round 6: what is serial -- the three state recurrences -- runs on THREE LANES
     * at once (lane 0 offsets, 1 match lengths, 2 literal lengths: the order their state bits are written in), 64 sequences
     * per batch: a lane's step is code -> (deltaNbBits, deltaFindState) -> bits out, next state; the bits a state gives up are
     * left in LDS.  Everything else is per sequence: lane j then assembles sequence j's up to 90 bits (three state fields, then
     * the literal-length, match-length and offset extra bits), a scan of the bit counts places them, and they are OR-ed into a
     * staging buffer in LDS whose whole dwords go out coalesced.  (Before: one sequence at a time with every value broadcast
     * from its lane and nine 64-bit shifts into a bit accumulator -- ~150 wave instructions per sequence, half of the entropy
     * stage.)  The LDS used lies over what this stage no longer needs: the code histogram (staging), the table-building cells
     * (the batch's codes) and the weights / normalised counts / cumulated counts (the states' bits). */
    {
        /* a block of many short matches at far offsets can cost more than it covers: the library's writer stops at the end of
         * its buffer and the block goes out raw; here nothing is written beyond the block's own size (+ the slack every output
         * slot has), and the size test below gives the same verdict */
        const uint32_t cap = src_size + 32u > op ? src_size + 32u - op : 0u;
        uint8_t *out = dst + op;
        uint32_t *stage = L.hist;                                    /* 256 dwords */
        uint8_t *bcodes = L.cell;                                    /* [64][3]: offset, match-length, literal-length code */
        uint16_t *recs = reinterpret_cast<uint16_t *>(L.wts);        /* [64][3]: bits out | count << 10 (wts + norm + cumul: 648 bytes) */
        static_assert(offsetof(EncLds, cell) - offsetof(EncLds, wts) >= 64u * 3u * 2u, "the states' bits need 384 bytes behind wts");
        for (uint32_t i = lane; i < 256u; i += 64u) stage[i] = 0;
        const FseCt *ct = lane == 0u ? &L.of : (lane == 1u ? &L.ml : &L.ll);
        uint32_t state = 0;       /* lanes 0..2 */
        uint32_t basebit = 0;     /* bit position of stage[0] in the stream (a multiple of 32) */
        uint32_t bits_total = 0;  /* bits produced so far */
        __builtin_amdgcn_wave_barrier();
        for (uint32_t c1 = nseq; c1 > 0u;) {
            const uint32_t cnt = c1 < 64u ? c1 : 64u;
            const uint32_t idx = lane < cnt ? c1 - 1u - lane : 0u; /* lane j holds sequence c1-1-j */
            const uint32_t lc = llc[idx], oc = ofc[idx], mc = mlc[idx];
            const uint2 q = seqs[idx];
            const uint32_t llv = q.y & 0xFFFFu, mlv = q.y >> 16, ofv = q.x;
            const uint32_t llb = kELLBits[lc], mlb = kEMLBits[mc];
            bcodes[lane * 3u + 0u] = (uint8_t)oc; bcodes[lane * 3u + 1u] = (uint8_t)mc; bcodes[lane * 3u + 2u] = (uint8_t)lc;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool first = c1 == nseq;
            if (lane < 3u) {
                uint32_t j = 0;
                if (first) { /* the last sequence starts the states and gives up no bits */
                    state = fse_init_state(*ct, bcodes[lane]);
                    recs[lane] = 0;
                    j = 1;
                }
                for (; j < cnt; j++) {
                    const uint32_t sym = bcodes[j * 3u + lane];
                    const uint32_t nb = (state + ct->dnb[sym]) >> 16;
                    recs[j * 3u + lane] = (uint16_t)((state & ((1u << nb) - 1u)) | (nb << 10));
                    state = ct->state[(int32_t)(state >> nb) + ct->dfind[sym]];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            /* lane j: sequence j's bits, least significant first: offset / match-length / literal-length state bits, then the
             * literal-length, match-length and offset extra bits */
            uint64_t lo = 0, hi = 0;
            uint32_t tb = 0;
            if (lane < cnt) {
                auto put = [&](uint32_t v, uint32_t nb) {
                    if (nb == 0u) return;
                    const uint64_t x = (uint64_t)v & ((1ull << nb) - 1ull);
                    if (tb < 64u) { lo |= x << tb; if (tb + nb > 64u) hi |= x >> (64u - tb); }
                    else hi |= x << (tb - 64u);
                    tb += nb;
                };
                const uint32_t r0 = recs[lane * 3u + 0u], r1 = recs[lane * 3u + 1u], r2 = recs[lane * 3u + 2u];
                put(r0 & 1023u, r0 >> 10); put(r1 & 1023u, r1 >> 10); put(r2 & 1023u, r2 >> 10);
                put(llv, llb); put(mlv, mlb); put(ofv, oc);
            }
            const uint32_t incl = scan64_incl(tb);
            const uint32_t rel = bits_total - basebit + (incl - tb); /* this lane's first bit inside the staging buffer */
            if (tb != 0u) {
                const uint32_t w = rel >> 5, sh = rel & 31u;
                const uint32_t d0 = (uint32_t)(lo << sh), d1 = (uint32_t)((lo >> 1) >> (31u - sh)), d2 = (uint32_t)(((lo >> 33) >> (31u - sh)) | (hi << sh)),
                               d3 = (uint32_t)((hi >> 1) >> (31u - sh));
                atomicOr(&stage[w], d0);
                if (d1) atomicOr(&stage[w + 1u], d1);
                if (d2) atomicOr(&stage[w + 2u], d2);
                if (d3) atomicOr(&stage[w + 3u], d3);
            }
            bits_total += lane_get(incl, 63);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            /* whole dwords out, the partial one to the front */
            const uint32_t ndw = (bits_total - basebit) >> 5;
            uint32_t carry = 0;
            for (uint32_t i = lane; i < ndw; i += 64u) {
                const uint32_t v = stage[i];
                const uint32_t bo = (basebit >> 3) + 4u * i;
                if (bo + 4u <= cap) __builtin_memcpy(out + bo, &v, 4);
            }
            carry = stage[ndw];
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            for (uint32_t i = lane; i <= ndw; i += 64u) stage[i] = i == 0u ? carry : 0u;
            if (ndw == 0u && lane == 0u) stage[0] = carry;
            basebit += 32u * ndw;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            c1 -= cnt;
        }
        /* the final states (match lengths, offsets, literal lengths), the end mark, the last bytes */
        {
            const uint32_t sm = lane_get(state, 1), so = lane_get(state, 0), sl = lane_get(state, 2);
            uint64_t acc = uni(stage[0]);
            uint32_t nacc = bits_total - basebit;
            auto add = [&](uint32_t v, uint32_t nb) { if (nb) { acc |= ((uint64_t)v & ((1ull << nb) - 1ull)) << nacc; nacc += nb; } };
            add(sm, (uint32_t)uni((uint32_t)L.ml.log)); add(so, (uint32_t)uni((uint32_t)L.of.log)); add(sl, (uint32_t)uni((uint32_t)L.ll.log));
            add(1u, 1u);
            const uint32_t nbytes = (nacc + 7u) >> 3, bo = basebit >> 3;
            if (lane < nbytes && bo + lane < cap) out[bo + lane] = (uint8_t)(acc >> (8u * lane));
            op += bo + nbytes;
        }
        zprof(hs, 7);
        if (last_ncount != 0xFFFFFFFFu && op - last_ncount < 4u) return 0;
    }
    if (op >= src_size - gain) return 0;
    return op;
}

/* ------------------------------------------------------------ match finder: strategy `fast` */
struct CPar { int wlog, clog, hlog, slog, mml, tlen, bt; /* bt: the binary-tree searcher (btlazy2) */ int ib; /* bits of a table index of this frame (block_fast_gbatch keeps a tag above them) */ };

__device__ inline uint32_t hash_ptr(const uint8_t *p, int hlog, int mls)
{
    switch (mls) {
    default:
    case 4: return (ld32u(p) * 2654435761u) >> (32 - hlog);
    case 5: return (uint32_t)(((ld64u(p) << 24) * 889523592379ull) >> (64 - hlog));
    case 6: return (uint32_t)(((ld64u(p) << 16) * 227718039650203ull) >> (64 - hlog));
    case 7: return (uint32_t)(((ld64u(p) << 8) * 58295818150454627ull) >> (64 - hlog));
    }
}

/* bytes equal from a/b forward, limited by end (64 bytes per step) */
__device__ inline uint32_t count_match(const uint8_t *a, const uint8_t *b, const uint8_t *end, uint32_t lane)
{
    {
        const bool inb = a + lane < end;
        const bool eq = inb && a[lane] == b[lane];
        const unsigned long long neq = __ballot(!eq);
        if (neq != 0ull) return ctz64(neq);
    }
    return count_long(a, b, end, lane, 64u);
}

struct SeqStore { uint32_t nseq, nlit, long_pos, long_kind; };

__device__ inline void store_seq(uint8_t *ws, SeqStore &ss, uint32_t ll, const uint8_t *lit, uint32_t offcode,
                                 uint32_t mlbase, uint32_t lane)
{
    uint8_t *lits = ws + kWsLit;
    for (uint32_t i = lane; i < ll; i += 64u) lits[ss.nlit + i] = lit[i];
    ss.nlit += ll;
    if (ll > 0xFFFFu) { ss.long_kind = 1; ss.long_pos = ss.nseq; }
    if (mlbase > 0xFFFFu) { ss.long_kind = 2; ss.long_pos = ss.nseq; }
    if (lane == 0) reinterpret_cast<uint2 *>(ws + kWsSeq)[ss.nseq] = make_uint2(offcode + 1u, (ll & 0xFFFFu) | (mlbase << 16));
    ss.nseq++;
}

/* ZSTD_compressBlock_fast; `table` is a generic pointer (LDS or global).  base = src - 1. */
__device__ uint32_t block_fast(uint32_t *table, const CPar &cp, const uint8_t *base, const uint8_t *istart,
                               uint32_t n, uint32_t *rep, uint8_t *ws, SeqStore &ss, uint32_t dict_limit,
                               uint32_t lane)
{
    const int hlog = cp.hlog, mls = cp.mml < 4 ? 4 : (cp.mml > 7 ? 7 : cp.mml);
    const uint32_t step_size = (uint32_t)cp.tlen + (cp.tlen ? 0u : 1u) + 1u;
    const uint8_t *ip0 = istart, *ip1, *anchor = istart;
    const uint32_t end_index = (uint32_t)(istart - base) + n;
    const uint32_t max_dist = 1u << cp.wlog;
    const uint32_t prefix_idx = (end_index - dict_limit > max_dist) ? end_index - max_dist : dict_limit;
    const uint8_t *prefix = base + prefix_idx;
    const uint8_t *iend = istart + n, *ilimit = iend - 8;
    uint32_t off1 = rep[0], off2 = rep[1], saved = 0;
    if (ip0 == prefix) ip0++;
    ip1 = ip0 + 1;
    {
        const uint32_t cur = (uint32_t)(ip0 - base);
        const uint32_t wlow = (cur - dict_limit > max_dist) ? cur - max_dist : dict_limit;
        const uint32_t max_rep = cur - wlow;
        if (off2 > max_rep) { saved = off2; off2 = 0; }
        if (off1 > max_rep) { saved = off1; off1 = 0; }
    }
    while (ip1 < ilimit) {
        uint32_t mlen, offcode;
        const uint8_t *ip2 = ip0 + 2;
        const uint32_t h0 = hash_ptr(ip0, hlog, mls), h1 = hash_ptr(ip1, hlog, mls);
        const uint32_t v0 = ld32u(ip0), v1 = ld32u(ip1);
        const uint32_t cur0 = (uint32_t)(ip0 - base), cur1 = (uint32_t)(ip1 - base);
        const uint32_t mi0 = uni(table[h0]);
        const uint32_t mi1 = uni(table[h1]);   /* read before either store, as the library does */
        const uint8_t *m0 = base + mi0, *m1 = base + mi1;
        table[h0] = cur0;
        table[h1] = cur1;
        bool have = false;
        if (off1 > 0u && ld32u(ip2 - off1) == ld32u(ip2)) {
            const uint8_t *rep_m = ip2 - off1;
            mlen = (uni(ip2[-1]) == uni(rep_m[-1])) ? 1u : 0u;
            ip0 = ip2 - mlen;
            m0 = rep_m - mlen;
            mlen += 4u;
            offcode = 0;
            have = true;
        } else {
            bool found = false;
            if (mi0 > prefix_idx && ld32u(m0) == v0) found = true;
            else if (mi1 > prefix_idx && ld32u(m1) == v1) { ip0 = ip1; m0 = m1; found = true; }
            if (found) {
                off2 = off1;
                off1 = (uint32_t)(ip0 - m0);
                offcode = off1 + 2u;
                mlen = 4u;
                while (ip0 > anchor && m0 > prefix && uni(ip0[-1]) == uni(m0[-1])) { ip0--; m0--; mlen++; }
                have = true;
            }
        }
        if (!have) {
            const uint32_t step = ((uint32_t)(ip0 - anchor) >> 7) + step_size;
            ip0 += step;
            ip1 += step;
            continue;
        }
        mlen += count_match(ip0 + mlen, m0 + mlen, iend, lane);
        store_seq(ws, ss, (uint32_t)(ip0 - anchor), anchor, offcode, mlen - 3u, lane);
        ip0 += mlen;
        anchor = ip0;
        if (ip0 <= ilimit) {
            table[hash_ptr(base + cur0 + 2u, hlog, mls)] = cur0 + 2u;
            table[hash_ptr(ip0 - 2, hlog, mls)] = (uint32_t)(ip0 - 2 - base);
            if (off2 > 0u) {
                while (ip0 <= ilimit && ld32u(ip0) == ld32u(ip0 - off2)) {
                    const uint32_t rlen = count_match(ip0 + 4, ip0 + 4 - off2, iend, lane) + 4u;
                    const uint32_t t = off2; off2 = off1; off1 = t;
                    table[hash_ptr(ip0, hlog, mls)] = (uint32_t)(ip0 - base);
                    ip0 += rlen;
                    store_seq(ws, ss, 0, anchor, 0, rlen - 3u, lane);
                    anchor = ip0;
                }
            }
        }
        ip1 = ip0 + 1;
    }
    rep[0] = off1 ? off1 : saved;
    rep[1] = off2 ? off2 : saved;
    return (uint32_t)(iend - anchor);
}


#include "zstd_dfast.h"
#include "zstd_lazy.h"
#include "zstd_opt.h"



} // namespace

/* One wave per frame, persistent grid.  The match finder's tables (u32, zeroed per frame) live in global memory
 * behind the workgroup's workspace: long/only table, then dfast's short table.  finder: 0 = `fast`, many
 * iterations per step (block_fast_gbatch); 1 = `dfast` (block_dfast_batch); 2 = `fast`, the serial walk
 * (block_fast: the plain restatement, CRYO_ZSTD_ENC=1); 3 = `greedy` (block_greedy: hash table, then chain table;
 * `width` carries searchLog). */
template <bool PROF, bool OPT> /* PROF: CRYO_ZSTD_STATS counters; OPT: the optimal-parser strategies (finder 7 btopt, 8 btultra, 9 btultra2) */
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4)))
k_zstd_enc(const uint8_t *__restrict__ src_base, uint64_t src_stride, uint32_t n, uint64_t n_blocks,
           uint8_t *__restrict__ dst_base, uint64_t dst_stride, int wlog, int hlog, int clog, int mml, int tlen,
           int finder, uint32_t width, uint32_t *__restrict__ out_size, int32_t *__restrict__ status,
           uint8_t *workspace, uint64_t ws_stride, unsigned long long *stats)
{
    __shared__ __attribute__((aligned(16))) EncLds L;
    /* the finders' mark array lies over the entropy stage's scratch (histogram + tree nodes: dead while a finder runs) */
    static_assert(offsetof(EncLds, nbyte) >= kDfMark, "mark array must stay inside the scratch part of EncLds");
    uint8_t *df_mark = reinterpret_cast<uint8_t *>(&L);
    unsigned long long t_mf = 0, t_en = 0, t_other = 0, t_prev = PROF ? __builtin_amdgcn_s_memtime() : 0; /* CRYO_ZSTD_STATS */
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t *ws = workspace + (uint64_t)blockIdx.x * ws_stride;
    CPar cp;
    cp.wlog = wlog; cp.clog = clog; cp.hlog = hlog; cp.slog = (int)width; cp.mml = mml; cp.tlen = tlen; cp.bt = finder >= 6 ? 1 : 0;
    cp.ib = 32 - __builtin_clz(n);
    const bool dfast = finder == 1;
    const bool two_tables = finder == 1 || finder >= 3;
    uint32_t *table = reinterpret_cast<uint32_t *>(ws + kWsBytes);
    uint32_t *tshort = table + (1u << hlog); /* dfast only */
    /* optimal parser: the 3-byte hash table (minMatch 3), the price table, the match ladder, the statistics between blocks */
    const int hlog3 = (OPT && mml == 3) ? (wlog < 17 ? wlog : 17) : 0;
    uint32_t *table3 = tshort + (1u << clog);
    OptT *opt_tab = reinterpret_cast<OptT *>(table3 + (hlog3 ? 1u << hlog3 : 0u));
    uint2 *opt_matches = reinterpret_cast<uint2 *>(reinterpret_cast<uint8_t *>(opt_tab) + kOptTabBytes);
    uint32_t *opt_saved = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(opt_matches) + kOptMatchBytes);

    for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint8_t *src = src_base + blk * src_stride;
        uint8_t *dst = dst_base + blk * dst_stride;
        {
            const uint32_t quads = ((1u << hlog) + (two_tables ? 1u << clog : 0u) + (hlog3 ? 1u << hlog3 : 0u)) / 4u;
            for (uint32_t i = lane; i < quads; i += 64u) reinterpret_cast<uint4 *>(table)[i] = make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t op;
        {
            const uint64_t wsize = 1ull << wlog;
            const uint32_t single = wsize >= n ? 1u : 0u;
            const uint32_t fcs = (n >= 256u) + (n >= 65536u + 256u);
            if (lane == 0) {
                dst[0] = 0x28; dst[1] = 0xB5; dst[2] = 0x2F; dst[3] = 0xFD;
                dst[4] = (uint8_t)((single << 5) + (fcs << 6));
            }
            op = 5;
            if (!single) { if (lane == 0) dst[op] = (uint8_t)((wlog - 10) << 3); op++; }
            if (fcs == 0u) { if (single) { if (lane == 0) dst[op] = (uint8_t)n; op++; } }
            else if (fcs == 1u) { if (lane == 0) { dst[op] = (uint8_t)(n - 256u); dst[op + 1] = (uint8_t)((n - 256u) >> 8); } op += 2; }
            else { if (lane == 0) { dst[op] = (uint8_t)n; dst[op + 1] = (uint8_t)(n >> 8); dst[op + 2] = (uint8_t)(n >> 16); dst[op + 3] = (uint8_t)(n >> 24); } op += 4; }
        }
        uint32_t rep[3] = {1, 4, 8};
        uint32_t dict_limit = 1;
        const uint8_t *base = src - 1; /* moves once for btultra2 (below) */
        bool first = true;
        OptStats ost;
        ost.f = reinterpret_cast<uint32_t *>(&L);
        ost.lit_sum = ost.ll_sum = ost.ml_sum = ost.of_sum = 0; /* a frame starts without statistics */
        ost.lit_base = ost.ll_base = ost.ml_base = ost.of_base = 0; ost.predef = false;
        HufState hs;
        hs.prev_valid = false; hs.next_new = false; hs.prof = PROF ? stats : nullptr; hs.t = 0; hs.strat = finder >= 3 ? (uint32_t)finder : (dfast ? 2u : 1u);
        HcState hc = {table, tshort, 1u, table3, hlog3, 1u};
        SeqTabs tb;
        tb.prev = reinterpret_cast<FseCt *>(ws + kWsTabs);
        tb.rep[0] = tb.rep[1] = tb.rep[2] = 0;
        uint32_t ip = 0;
        while (ip < n) {
            const uint32_t bs = (n - ip < kZBlk) ? n - ip : kZBlk;
            const uint32_t last = (ip + bs == n) ? 1u : 0u;
            uint32_t csize = 0;
            /* window.dictLimit stays 1 for the whole frame (libzstd 1.4.8: ZSTD_compress_frameChunk only checks
             * dictionary validity); the window is enforced per block through the lowest-prefix index */
            if (bs >= 7u) {
                SeqStore ss;
                ss.nseq = 0; ss.nlit = 0; ss.long_pos = 0; ss.long_kind = 0;
                uint32_t nrep[3] = {rep[0], rep[1], rep[2]};
                if constexpr (PROF) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_other += t - t_prev; t_prev = t; }
                uint32_t last_ll;
                if constexpr (OPT) {
                    const uint32_t cur = (uint32_t)(src + ip - base); /* ZSTD_buildSeqStore: limited catch-up after a very long match */
                    if (cur > hc.next_to_update + 384u) {
                        const uint32_t d = cur - hc.next_to_update - 384u;
                        hc.next_to_update = cur - (d < 192u ? d : 192u);
                    }
                    if (!first) { /* the statistics of the previous blocks come back into LDS */
                        for (uint32_t i = lane; i < kOsWords; i += 64u) ost.f[i] = opt_saved[i];
                        lds_sync();
                    }
                    if (finder == 9 && first && bs > 1024u) {
                        /* ZSTD_compressBlock_btultra2 -> ZSTD_initStats_ultra: the first block of a frame is parsed twice.  The
                         * first pass only collects statistics; it is then forgotten by moving the window base, so that every
                         * index it left in the tables lies below the lowest valid one */
                        uint32_t trep[3] = {rep[0], rep[1], rep[2]};
                        (void)block_opt(hc, cp, ost, opt_tab, opt_matches, base, src + ip, bs, trep, ws, ss, 2, lane);
                        ss.nseq = 0; ss.nlit = 0; ss.long_pos = 0; ss.long_kind = 0;
                        base -= bs;
                        hc.low += bs;
                        hc.next_to_update = hc.low;
                        ost.lit_sum = opt_upscale(ost.f + kOsLit, 255u, lane);
                        ost.ll_sum = opt_upscale(ost.f + kOsLL, kMaxLL, lane);
                        ost.ml_sum = opt_upscale(ost.f + kOsML, kMaxML, lane);
                        ost.of_sum = opt_upscale(ost.f + kOsOF, kMaxOff, lane);
                    }
                    last_ll = block_opt(hc, cp, ost, opt_tab, opt_matches, base, src + ip, bs, nrep, ws, ss, finder == 7 ? 0 : 2, lane);
                    for (uint32_t i = lane; i < kOsWords; i += 64u) opt_saved[i] = ost.f[i];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                }
                else if (finder == 1) last_ll = block_dfast_batch<PROF>(table, tshort, df_mark, cp, base, src + ip, bs, nrep, ws, ss, dict_limit, lane, width, stats);
                else if (finder >= 3) {
                    const uint32_t cur = ip + 1u; /* ZSTD_buildSeqStore: limited catch-up after a very long match */
                    if (cur > hc.next_to_update + 384u) {
                        const uint32_t d = cur - hc.next_to_update - 384u;
                        hc.next_to_update = cur - (d < 192u ? d : 192u);
                    }
                    last_ll = block_lazy(hc, df_mark, cp, finder == 6 ? 2 : finder - 3, base, src + ip, bs, nrep, ws, ss, lane);
                }
                else if (finder == 0) last_ll = block_fast_gbatch(table, df_mark, cp, base, src + ip, bs, nrep, ws, ss, dict_limit, lane, width);
                else last_ll = block_fast(table, cp, base, src + ip, bs, nrep, ws, ss, dict_limit, lane);
                if constexpr (PROF) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_mf += t - t_prev; t_prev = t; }
                wave_copy_bytes(ws + kWsLit + ss.nlit, src + ip + bs - last_ll, last_ll, lane);
                ss.nlit += last_ll;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if constexpr (PROF) hs.t = __builtin_amdgcn_s_memtime();
                csize = compress_sequences(L, dst + op + 3, reinterpret_cast<const uint2 *>(ws + kWsSeq), ws + kWsLit, ws + kWsLlc, ss.nseq, ss.nlit, bs, ss.long_pos, ss.long_kind, hs, tb,
                                           (finder == 0 || finder == 2) && tlen > 0 /* literals stay raw only for `fast` with a target length */, lane);
                if constexpr (PROF) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_en += t - t_prev; t_prev = t; }
                if (!first && csize < 25u) { /* RLE block for constant non-first blocks */
                    const uint32_t b0 = uni(src[ip]);
                    bool diff = false;
                    {
                        const uint32_t b4 = b0 * 0x01010101u;
                        uint32_t o = 0;
                        for (; o + 1024u <= bs; o += 1024u) {
                            uint4 v;
                            __builtin_memcpy(&v, src + ip + o + 16u * lane, 16);
                            diff |= ((v.x ^ b4) | (v.y ^ b4) | (v.z ^ b4) | (v.w ^ b4)) != 0u;
                        }
                        for (uint32_t i = o + lane; i < bs; i += 64u) diff |= (src[ip + i] != b0);
                    }
                    if (__ballot(diff) == 0ull) { csize = 1; if (lane == 0) dst[op + 3] = (uint8_t)b0; }
                }
                if (csize > 1u) {
                    rep[0] = nrep[0]; rep[1] = nrep[1]; rep[2] = nrep[2];
                    if (hs.next_new) {
                        for (uint32_t i = lane; i < 256u; i += 64u) { L.pval[i] = L.hval[i]; L.pnb[i] = L.hnb[i]; }
                        hs.prev_valid = true;
                    }
                    if (finder >= 4 && !last && tb.built) { /* ZSTD_confirmRepcodesAndEntropyTables: the sequence tables too */
                        tb.rep[0] = tb.nrep[0]; tb.rep[1] = tb.nrep[1]; tb.rep[2] = tb.nrep[2];
                        const uint32_t *t0 = reinterpret_cast<const uint32_t *>(&L.ll), *t1 = reinterpret_cast<const uint32_t *>(&L.of),
                                       *t2 = reinterpret_cast<const uint32_t *>(&L.ml);
                        uint32_t *p = reinterpret_cast<uint32_t *>(tb.prev);
                        constexpr uint32_t W = sizeof(FseCt) / 4u;
                        for (uint32_t i = lane; i < W; i += 64u) { p[i] = t0[i]; p[W + i] = t1[i]; p[2u * W + i] = t2[i]; }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    }
                }
            }
            if (csize == 0u) {
                const uint32_t h = last + (0u << 1) + (bs << 3);
                if (lane == 0) { dst[op] = (uint8_t)h; dst[op + 1] = (uint8_t)(h >> 8); dst[op + 2] = (uint8_t)(h >> 16); }
                wave_copy_bytes(dst + op + 3u, src + ip, bs, lane);
                op += 3u + bs;
            } else {
                const uint32_t h = csize == 1u ? last + (1u << 1) + (bs << 3) : last + (2u << 1) + (csize << 3);
                if (lane == 0) { dst[op] = (uint8_t)h; dst[op + 1] = (uint8_t)(h >> 8); dst[op + 2] = (uint8_t)(h >> 16); }
                op += 3u + csize;
            }
            ip += bs;
            first = false;
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) { out_size[blk] = op; status[blk] = CRYO_ST_OK; }
        __builtin_amdgcn_wave_barrier();
    }
    if (PROF && stats && lane == 0) {
        t_other += __builtin_amdgcn_s_memtime() - t_prev;
        atomicAdd(&stats[0], t_mf); atomicAdd(&stats[1], t_en); atomicAdd(&stats[2], t_other);
    }
}

/* ZSTD_getCParams + ZSTD_adjustCParams for the levels with a kernel at cryo block sizes (oracle-checked): `fast`
 * (-5..2), `dfast` (3, 4), `greedy` (5; 6 above 256 KiB), `lazy` / `lazy2`, `btlazy2`, `btopt` / `btultra` / `btultra2` (up to
 * 22).  *clog: dfast's short table / the chain table / the binary tree; *strategy: 1 fast, 2 dfast, 3 greedy, 4 lazy, 5 lazy2,
 * 6 btlazy2, 7 btopt, 8 btultra, 9 btultra2; *slog: searchLog. */
static bool zstd_fast_cparams(int level, uint32_t n, int *wlog, int *hlog, int *mml, int *tlen, int *clog = nullptr,
                              bool *dfast = nullptr, int *strategy = nullptr, int *slog = nullptr)
{
    /* libzstd 1.4.8's four parameter tables (ZSTD_defaultCParameters: source size > 256 KiB, <= 256 KiB, <= 128 KiB,
     * <= 16 KiB), rows: the base row of the negative levels, then levels 1 .. 12; columns: windowLog, chainLog, hashLog,
     * searchLog, minMatch, targetLength, strategy (1 fast, 2 dfast, 3 greedy, 4 lazy, 5 lazy2, 6 btlazy2, 7 btopt, 8 btultra,
     * 9 btultra2).  Dumped from ZSTD_getCParams and checked against it by the tests. */
    static const int kCParTab[4][23][7] = {
        {{19, 12, 13, 1, 6, 1, 1}, {19, 13, 14, 1, 7, 0, 1}, {20, 15, 16, 1, 6, 0, 1}, {21, 16, 17, 1, 5, 0, 2}, {21, 18, 18, 1, 5, 0, 2}, {21, 18, 19, 2, 5, 2, 3}, {21, 19, 19, 3, 5, 4, 3}, {21, 19, 19, 3, 5, 8, 4}, {21, 19, 19, 3, 5, 16, 5}, {21, 19, 20, 4, 5, 16, 5}, {22, 20, 21, 4, 5, 16, 5}, {22, 21, 22, 4, 5, 16, 5}, {22, 21, 22, 5, 5, 16, 5}, {22, 21, 22, 5, 5, 32, 6}, {22, 22, 23, 5, 5, 32, 6}, {22, 23, 23, 6, 5, 32, 6}, {22, 22, 22, 5, 5, 48, 7}, {23, 23, 22, 5, 4, 64, 7}, {23, 23, 22, 6, 3, 64, 8}, {23, 24, 22, 7, 3, 256, 9}, {25, 25, 23, 7, 3, 256, 9}, {26, 26, 24, 7, 3, 512, 9}, {27, 27, 25, 9, 3, 999, 9}},
        {{18, 12, 13, 1, 5, 1, 1}, {18, 13, 14, 1, 6, 0, 1}, {18, 14, 14, 1, 5, 0, 2}, {18, 16, 16, 1, 4, 0, 2}, {18, 16, 17, 2, 5, 2, 3}, {18, 18, 18, 3, 5, 2, 3}, {18, 18, 19, 3, 5, 4, 4}, {18, 18, 19, 4, 4, 4, 4}, {18, 18, 19, 4, 4, 8, 5}, {18, 18, 19, 5, 4, 8, 5}, {18, 18, 19, 6, 4, 8, 5}, {18, 18, 19, 5, 4, 12, 6}, {18, 19, 19, 7, 4, 12, 6}, {18, 18, 19, 4, 4, 16, 7}, {18, 18, 19, 4, 3, 32, 7}, {18, 18, 19, 6, 3, 128, 7}, {18, 19, 19, 6, 3, 128, 8}, {18, 19, 19, 8, 3, 256, 8}, {18, 19, 19, 6, 3, 128, 9}, {18, 19, 19, 8, 3, 256, 9}, {18, 19, 19, 10, 3, 512, 9}, {18, 19, 19, 12, 3, 512, 9}, {18, 19, 19, 13, 3, 999, 9}},
        {{17, 12, 12, 1, 5, 1, 1}, {17, 12, 13, 1, 6, 0, 1}, {17, 13, 15, 1, 5, 0, 1}, {17, 15, 16, 2, 5, 0, 2}, {17, 17, 17, 2, 4, 0, 2}, {17, 16, 17, 3, 4, 2, 3}, {17, 17, 17, 3, 4, 4, 4}, {17, 17, 17, 3, 4, 8, 5}, {17, 17, 17, 4, 4, 8, 5}, {17, 17, 17, 5, 4, 8, 5}, {17, 17, 17, 6, 4, 8, 5}, {17, 17, 17, 5, 4, 8, 6}, {17, 18, 17, 7, 4, 12, 6}, {17, 18, 17, 3, 4, 12, 7}, {17, 18, 17, 4, 3, 32, 7}, {17, 18, 17, 6, 3, 256, 7}, {17, 18, 17, 6, 3, 128, 8}, {17, 18, 17, 8, 3, 256, 8}, {17, 18, 17, 10, 3, 512, 8}, {17, 18, 17, 5, 3, 256, 9}, {17, 18, 17, 7, 3, 512, 9}, {17, 18, 17, 9, 3, 512, 9}, {17, 18, 17, 11, 3, 999, 9}},
        {{14, 12, 13, 1, 5, 1, 1}, {14, 14, 15, 1, 5, 0, 1}, {14, 14, 15, 1, 4, 0, 1}, {14, 14, 15, 2, 4, 0, 2}, {14, 14, 14, 4, 4, 2, 3}, {14, 14, 14, 3, 4, 4, 4}, {14, 14, 14, 4, 4, 8, 5}, {14, 14, 14, 6, 4, 8, 5}, {14, 14, 14, 8, 4, 8, 5}, {14, 15, 14, 5, 4, 8, 6}, {14, 15, 14, 9, 4, 8, 6}, {14, 15, 14, 3, 4, 12, 7}, {14, 15, 14, 4, 3, 24, 7}, {14, 15, 14, 5, 3, 32, 8}, {14, 15, 15, 6, 3, 64, 8}, {14, 15, 15, 7, 3, 256, 8}, {14, 15, 15, 5, 3, 48, 9}, {14, 15, 15, 6, 3, 128, 9}, {14, 15, 15, 7, 3, 256, 9}, {14, 15, 15, 8, 3, 256, 9}, {14, 15, 15, 8, 3, 512, 9}, {14, 15, 15, 9, 3, 512, 9}, {14, 15, 15, 10, 3, 999, 9}}};
    const int (*t)[7];
    if (level == 0) level = 3;
    if (level > 22 || level < -131072 || n == 0u) return false;
    t = kCParTab[n > 256u * 1024u ? 0 : (n > 128u * 1024u ? 1 : (n > 16u * 1024u ? 2 : 3))];
    const int row = level < 0 ? 0 : level;
    int cl = t[row][1];
    *wlog = t[row][0]; *hlog = t[row][2]; *mml = t[row][4];
    *tlen = level < 0 ? -level : t[row][5];
    int srclog = 0;
    for (uint32_t v = n - 1u; v; v >>= 1) srclog++;
    if (n < 64u) srclog = 6;
    if (*wlog > srclog) *wlog = srclog;
    if (*hlog > *wlog + 1) *hlog = *wlog + 1;
    {
        const int btscale = t[row][6] >= 6 ? 1 : 0; /* cycleLog = chainLog - 1 for the binary tree */
        if (cl - btscale > *wlog) cl = *wlog + btscale;
    }
    if (*wlog < 10) *wlog = 10;
    if (clog) *clog = cl;
    if (dfast) *dfast = t[row][6] == 2;
    if (strategy) *strategy = t[row][6];
    if (slog) *slog = t[row][3];
    return true;
}

/* workgroups per CU: LDS (entropy-stage tables; the finders' mark array lies inside) admits 11, the registers 12 */
static uint32_t zstd_enc_grid(uint64_t n_blocks, size_t stride)
{
    uint64_t per_cu = (160u * 1024u) / (sizeof(EncLds) + 64u);
    if (per_cu > 16) per_cu = 16;
    static const uint64_t grid_env = cryo_tuning_env("CRYO_ZSTD_ENC_GRID") ? (uint64_t)atoll(cryo_tuning_env("CRYO_ZSTD_ENC_GRID")) : 0; /* tuning aid */
    uint64_t cap = grid_env ? grid_env : 256u * per_cu;
    /* the deep levels' tables reach 12 MiB per workgroup: keep the workspace under 24 GiB, at least one workgroup per CU */
    const uint64_t fit = ((uint64_t)24 << 30) / stride;
    if (cap > fit) cap = fit < 256u ? 256u : fit;
    return (uint32_t)(n_blocks < cap ? n_blocks : cap);
}
/* per workgroup: sequences, literals, codes (kWsBytes), then the match finder's table(s) */
static size_t zstd_enc_stride(int hlog, int clog, bool two_tables, int strategy = 0, int mml = 0, int wlog = 0)
{
    size_t b = kWsBytes + (((size_t)4u << hlog) + (two_tables ? (size_t)4u << clog : 0u));
    if (strategy >= 7) b += (mml == 3 ? (size_t)4u << (wlog < 17 ? wlog : 17) : 0u) + kOptExtraBytes; /* zstd_opt.h */
    return b;
}

size_t zstd_compress_workspace(uint64_t n_blocks, int level, uint32_t block_size)
{
    int wlog, hlog, mml, tlen, clog;
    bool dfast = false;
    int strategy = 1;
    if (!zstd_fast_cparams(level, block_size, &wlog, &hlog, &mml, &tlen, &clog, &dfast, &strategy)) return 256;
    const size_t stride = zstd_enc_stride(hlog, clog, strategy >= 2, strategy, mml, wlog);
    return (size_t)zstd_enc_grid(n_blocks, stride) * stride + 256;
}

bool zstd_compress_supported(int level, uint32_t block_size)
{
    int a, b, c, d;
    return zstd_fast_cparams(level, block_size, &a, &b, &c, &d);
}

hipError_t launch_zstd_compress(hipStream_t s, const uint8_t *d_src, uint64_t src_stride, uint32_t block_size,
                                uint64_t n_blocks, uint8_t *d_dst, uint64_t dst_stride, int level,
                                uint32_t *d_out_size, int32_t *d_status, void *d_workspace, size_t workspace_bytes)
{
    if (n_blocks == 0) return hipSuccess;
    int wlog, hlog, mml, tlen, clog;
    bool dfast = false;
    int strategy = 1, slog = 0;
    if (!zstd_fast_cparams(level, block_size, &wlog, &hlog, &mml, &tlen, &clog, &dfast, &strategy, &slog)) return hipErrorNotSupported;
    static const bool serial_only = cryo_tuning_env("CRYO_ZSTD_ENC") && cryo_tuning_env("CRYO_ZSTD_ENC")[0] == '1'; /* testing aid: the serial `fast` walk */
    const int finder = strategy >= 3 ? strategy : (dfast ? 1 : (serial_only ? 2 : 0)); /* 3 greedy, 4 lazy, 5 lazy2, 6 btlazy2 */
    /* search positions (dfast) / iterations (fast: two positions each) per step.  Measured on text-like rows, GB/s:
     * dfast level 3  16: 6.6  32: 7.4  64: 6.9;  fast level 1  16: 14.4  32: 13.9  64: 13.1 -- wider steps read
     * table slots for positions behind the first match, narrower ones pay more trips per sequence */
    static const uint32_t w_env = cryo_tuning_env("CRYO_ZSTD_ENC_WIDTH") ? (uint32_t)atoi(cryo_tuning_env("CRYO_ZSTD_ENC_WIDTH")) : 0u; /* tuning aid */
    const uint32_t width = strategy >= 3 ? (uint32_t)slog : (w_env ? w_env : (dfast ? 32u : 16u));
    const size_t stride = zstd_enc_stride(hlog, clog, strategy >= 2, strategy, mml, wlog);
    uint32_t grid = zstd_enc_grid(n_blocks, stride);
    if (workspace_bytes < (size_t)grid * stride) return hipErrorInvalidValue;
    static const bool stats_env = cryo_tuning_env("CRYO_ZSTD_STATS") != nullptr; /* debugging aid */
    const bool want_stats = stats_env && strategy < 7;
    unsigned long long *d_st = nullptr, h_st[24] = {0};
    if (want_stats) {
        if (hipMalloc((void **)&d_st, sizeof h_st) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d_st, 0, sizeof h_st, s);
    }
    /* one kernel for all finders, chosen at run time: instantiating it per finder makes the compiler inline each
     * finder into the frame loop and spill three times as much */
    if (strategy >= 7)
        hipLaunchKernelGGL((k_zstd_enc<false, true>), dim3(grid), dim3(64), 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride, wlog,
                           hlog, clog, mml, tlen, finder, width, d_out_size, d_status, (uint8_t *)d_workspace, (uint64_t)stride, d_st);
    else if (want_stats)
        hipLaunchKernelGGL((k_zstd_enc<true, false>), dim3(grid), dim3(64), 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride, wlog,
                           hlog, clog, mml, tlen, finder, width, d_out_size, d_status, (uint8_t *)d_workspace, (uint64_t)stride, d_st);
    else
        hipLaunchKernelGGL((k_zstd_enc<false, false>), dim3(grid), dim3(64), 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride, wlog,
                           hlog, clog, mml, tlen, finder, width, d_out_size, d_status, (uint8_t *)d_workspace, (uint64_t)stride, d_st);
    if (want_stats) {
        (void)hipMemcpyAsync(h_st, d_st, sizeof h_st, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        (void)hipFree(d_st);
        const double tot = (double)(h_st[0] + h_st[1] + h_st[2]);
        fprintf(stderr, "[zstd enc cycles] match finder %.1f%%  entropy stage %.1f%%  other %.1f%%\n",
                100.0 * h_st[0] / tot, 100.0 * h_st[1] / tot, 100.0 * h_st[2] / tot);
        fprintf(stderr, "[zstd enc cycles] of the entropy stage: literal histogram %.1f%%  huffman build+table %.1f%%  "
                        "huffman encode %.1f%%  sequence codes+tables %.1f%%  FSE encode %.1f%%\n",
                100.0 * h_st[3] / h_st[1], 100.0 * h_st[4] / h_st[1], 100.0 * h_st[5] / h_st[1], 100.0 * h_st[6] / h_st[1],
                100.0 * h_st[7] / h_st[1]);
        if (dfast) {
            double t = 0;
            for (int k = 8; k < 16; k++) t += (double)h_st[k];
            fprintf(stderr, "[zstd enc cycles] dfast finder: input+hash %.1f%%  dup filter %.1f%%  table gather %.1f%%  candidates %.1f%%  "
                            "match (count, lookups) %.1f%%  tail (insertions, repeats, next input) %.1f%% | steps %llu, positions/step %.1f, "
                            "steps/sequence %.2f, cycles/sequence %.0f; tails from the window %llu, immediate repeats seen there %llu, repeat sequences %llu\n",
                    100.0 * h_st[8] / t, 100.0 * h_st[9] / t, 100.0 * h_st[10] / t, 100.0 * h_st[11] / t, 100.0 * h_st[12] / t,
                    100.0 * h_st[14] / t, h_st[16], (double)h_st[17] / h_st[16], (double)h_st[16] / h_st[18], t / h_st[18], h_st[19], h_st[20], h_st[21]);
        }
    }
    return hipGetLastError();
}

} // namespace cryo
