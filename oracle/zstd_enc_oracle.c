/*
 * zstd_enc_oracle.c -- CPU ORACLE (test infrastructure, see cryo_oracle.h).
 *
 * Restates ZSTD_compress(dst, ZSTD_compressBound(B), src, B, level) as libzstd 1.4.8 runs it
 * for the reference's call shape (compression.c:102-104), at every level (-5 .. 22): the strategies `fast` (levels -5 .. 2 at
 * cryo block sizes; the reference's default level 1 is one of them), `dfast`, `greedy`, `lazy`, `lazy2`,
 * `btlazy2`, and the optimal parsers `btopt`, `btultra`, `btultra2`.  Output bytes are identical to the library's; pinned by
 * tests/golden/vectors.json (libzstd 1.4.8 == 1.4.9) and by live differential tests on every size class.
 *
 * Pipeline restated (all integer arithmetic):
 *   parameters by level and size -> frame header -> per 128 KiB block:
 *     greedy 2-position hash-table match finder with repeat-offset checks ("fast" strategy), or
 *     the two-table (8-byte long hash + short hash) finder of the "dfast" strategy, or the hash-chain
 *     searcher with the greedy / lazy (depth 1, 2) parsers, or the binary tree with the lazy2 parser or the optimal parser
 *     (prices from running symbol statistics, cheapest path over up to 4096 positions)
 *     -> sequences (literal length, match length, offset code) + literal bytes
 *     -> literals: raw / RLE / Huffman (length-limited tree, FSE-compressed or raw weights,
 *        1 or 4 backward bitstreams), with the library's "worth it" heuristics
 *     -> sequences: per-field encoding type (predefined / RLE / FSE / repeat of the previous block's
 *        table) by thresholds below `lazy`, by estimated costs from `lazy` on, FSE table normalisation + description, interleaved backward bitstream
 *     -> raw block fallback when the gain is below srcSize/64 + 2 (/128, /256 for btultra, btultra2); RLE block for constant
 *        non-first blocks
 */
#include "cryo_oracle.h"
#include <string.h>
#include <math.h>

#define ZBLOCK_MAX (128u * 1024u)
#define MINMATCH 3
#define REP_MOVE 2
#define HASH_READ 8
#define MaxLL 35
#define MaxML 52
#define MaxOff 31
#define DefaultMaxOff 28
#define LLFSELog 9
#define MLFSELog 9
#define OffFSELog 8
#define HUF_LOG_MAX 12
#define HUF_LOG_DEFAULT 11

static int hb(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* ------------------------------------------------------------ forward LSB-first bit writer */
typedef struct { uint8_t *p; uint64_t acc; int n; size_t len; size_t cap; /* bytes that may be written; len counts on */ } bitw;
static void bw_init(bitw *b, uint8_t *p) { b->p = p; b->acc = 0; b->n = 0; b->len = 0; b->cap = (size_t)-1; }
static void bw_add(bitw *b, uint64_t v, int nb)
{
    if (nb == 0) return;
    b->acc |= (v & ((nb >= 64) ? ~0ull : ((1ull << nb) - 1))) << b->n;
    b->n += nb;
    while (b->n >= 8) { if (b->len < b->cap) b->p[b->len] = (uint8_t)b->acc; b->len++; b->acc >>= 8; b->n -= 8; }
}
/* end mark + padding, as BIT_closeCStream */
static size_t bw_close(bitw *b)
{
    bw_add(b, 1, 1);
    if (b->n > 0) { if (b->len < b->cap) b->p[b->len] = (uint8_t)b->acc; b->len++; b->n = 0; b->acc = 0; }
    return b->len;
}
/* flush without end mark (FSE table descriptions) */
static size_t bw_flush(bitw *b)
{
    if (b->n > 0) { if (b->len < b->cap) b->p[b->len] = (uint8_t)b->acc; b->len++; b->n = 0; b->acc = 0; }
    return b->len;
}

/* ------------------------------------------------------------ FSE (compression side) */
typedef struct { int delta_find; uint32_t delta_nb; } fse_sym;
typedef struct {
    int log;
    uint32_t max_sym;         /* maxSymbolValue of the table (repeat-mode cost test) */
    uint16_t state[1 << 9];   /* table sizes here are <= 512 */
    fse_sym sym[256];
} fse_ct;

static int fse_min_log(size_t n, uint32_t max_sym)
{
    int a = hb((uint32_t)n) + 1, b = hb(max_sym) + 2;
    return a < b ? a : b;
}
static int fse_optimal_log(int max_log, size_t n, uint32_t max_sym, int minus)
{
    /* FSE_optimalTableLog_internal computes highbit32(srcSize - 1) - minus in unsigned arithmetic: below 2^minus + 1 symbols it
     * wraps and limits nothing (found by the soak on a block of four sequences: the library kept tableLog 8, priced a new
     * offset table higher than the predefined one, and chose the latter) */
    int max_src = hb((uint32_t)(n - 1)) - minus;
    int log = max_log, min_bits = fse_min_log(n, max_sym);
    if (log == 0) log = 11;
    if (max_src >= 0 && max_src < log) log = max_src;
    if (min_bits > log) log = min_bits;
    if (log < 5) log = 5;
    if (log > 12) log = 12;
    return log;
}

/* secondary normalisation (rare corner case of FSE_normalizeCount) */
static int fse_norm_m2(int16_t *norm, int log, const uint32_t *count, size_t total, uint32_t max_sym, int16_t low_prob)
{
    const int16_t NYA = -2;
    uint32_t s, distributed = 0, to_dist;
    const uint32_t low_thr = (uint32_t)(total >> log);
    uint32_t low_one = (uint32_t)((total * 3) >> (log + 1));
    for (s = 0; s <= max_sym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= low_thr) { norm[s] = low_prob; distributed++; total -= count[s]; continue; }
        if (count[s] <= low_one) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = NYA;
    }
    to_dist = (1u << log) - distributed;
    if (to_dist == 0) return 0;
    if ((total / to_dist) > low_one) {
        low_one = (uint32_t)((total * 3) / (to_dist * 2));
        for (s = 0; s <= max_sym; s++)
            if (norm[s] == NYA && count[s] <= low_one) { norm[s] = 1; distributed++; total -= count[s]; }
        to_dist = (1u << log) - distributed;
    }
    if (distributed == max_sym + 1) {
        uint32_t mv = 0, mc = 0;
        for (s = 0; s <= max_sym; s++) if (count[s] > mc) { mv = s; mc = count[s]; }
        norm[mv] += (int16_t)to_dist;
        return 0;
    }
    if (total == 0) {
        for (s = 0; to_dist > 0; s = (s + 1) % (max_sym + 1)) if (norm[s] > 0) { to_dist--; norm[s]++; }
        return 0;
    }
    {
        const uint64_t vlog = 62 - (uint64_t)log, mid = (1ull << (vlog - 1)) - 1;
        const uint64_t rstep = (((1ull << vlog) * to_dist) + mid) / total;
        uint64_t tmp = mid;
        for (s = 0; s <= max_sym; s++) {
            if (norm[s] == NYA) {
                const uint64_t end = tmp + (uint64_t)count[s] * rstep;
                const uint32_t w = (uint32_t)(end >> vlog) - (uint32_t)(tmp >> vlog);
                if (w < 1) return -1;
                norm[s] = (int16_t)w;
                tmp = end;
            }
        }
    }
    return 0;
}

/* FSE_normalizeCount; returns table log, 0 for the RLE special case, <0 on error.
 * use_low_prob: symbols at or below total >> log get the "less than one" probability -1
 * (sequence tables with >= 2048 sequences) or a plain 1 (Huffman weights, short sequence tables) */
static int fse_normalize(int16_t *norm, int log, const uint32_t *count, size_t total, uint32_t max_sym, int use_low_prob)
{
    const int16_t low_prob = use_low_prob ? -1 : 1;
    static const uint32_t rtb[] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
    const uint64_t scale = 62 - (uint64_t)log, step = (1ull << 62) / total, vstep = 1ull << (scale - 20);
    int still = 1 << log;
    uint32_t s, largest = 0;
    int16_t largest_p = 0;
    const uint32_t low_thr = (uint32_t)(total >> log);
    if (log < fse_min_log(total, max_sym)) return -1;
    for (s = 0; s <= max_sym; s++) {
        if (count[s] == total) return 0;
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= low_thr) { norm[s] = low_prob; still--; }
        else {
            int16_t proba = (int16_t)(((uint64_t)count[s] * step) >> scale);
            if (proba < 8) {
                const uint64_t rest = vstep * rtb[proba];
                proba += ((uint64_t)count[s] * step) - ((uint64_t)proba << scale) > rest;
            }
            if (proba > largest_p) { largest_p = proba; largest = s; }
            norm[s] = proba;
            still -= proba;
        }
    }
    if (-still >= (norm[largest] >> 1)) { if (fse_norm_m2(norm, log, count, total, max_sym, low_prob)) return -1; }
    else norm[largest] += (int16_t)still;
    return log;
}

/* FSE_writeNCount; returns bytes written */
static size_t fse_write_ncount(uint8_t *dst, const int16_t *norm, uint32_t max_sym, int log)
{
    bitw b;
    const int table_size = 1 << log;
    int remaining = table_size + 1, threshold = table_size, nb = log + 1, prev0 = 0;
    uint32_t sym = 0;
    const uint32_t alpha = max_sym + 1;
    bw_init(&b, dst);
    bw_add(&b, (uint64_t)(log - 5), 4);
    while (sym < alpha && remaining > 1) {
        if (prev0) {
            uint32_t start = sym;
            while (sym < alpha && !norm[sym]) sym++;
            if (sym == alpha) break;
            while (sym >= start + 24) { start += 24; bw_add(&b, 0xFFFF, 16); }
            while (sym >= start + 3) { start += 3; bw_add(&b, 3, 2); }
            bw_add(&b, sym - start, 2);
        }
        {
            int count = norm[sym++];
            const int max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bw_add(&b, (uint64_t)count, nb - (count < max));
            prev0 = (count == 1);
            if (remaining < 1) return 0;
            while (remaining < threshold) { nb--; threshold >>= 1; }
        }
    }
    if (remaining != 1) return 0;
    return bw_flush(&b);
}

static void fse_build_ct(fse_ct *ct, const int16_t *norm, uint32_t max_sym, int log)
{
    const uint32_t size = 1u << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint8_t cell[1 << 9];
    uint32_t cumul[258];
    uint32_t high = size - 1, pos = 0, u;
    ct->log = log;
    ct->max_sym = max_sym;
    cumul[0] = 0;
    for (u = 1; u <= max_sym + 1; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; cell[high--] = (uint8_t)(u - 1); }
        else cumul[u] = cumul[u - 1] + (uint32_t)norm[u - 1];
    }
    cumul[max_sym + 1] = size + 1;
    for (u = 0; u <= max_sym; u++) {
        int i;
        for (i = 0; i < norm[u]; i++) {
            cell[pos] = (uint8_t)u;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    for (u = 0; u < size; u++) { const uint8_t s = cell[u]; ct->state[cumul[s]++] = (uint16_t)(size + u); }
    {
        uint32_t total = 0, s;
        for (s = 0; s <= max_sym; s++) {
            switch (norm[s]) {
            case 0: ct->sym[s].delta_nb = ((uint32_t)(log + 1) << 16) - (1u << log); ct->sym[s].delta_find = 0; break;
            case -1:
            case 1:
                ct->sym[s].delta_nb = ((uint32_t)log << 16) - (1u << log);
                ct->sym[s].delta_find = (int)total - 1;
                total++;
                break;
            default: {
                const uint32_t max_out = (uint32_t)log - (uint32_t)hb((uint32_t)norm[s] - 1);
                const uint32_t min_plus = (uint32_t)norm[s] << max_out;
                ct->sym[s].delta_nb = (max_out << 16) - min_plus;
                ct->sym[s].delta_find = (int)total - norm[s];
                total += (uint32_t)norm[s];
            }
            }
        }
    }
}
static void fse_build_ct_rle(fse_ct *ct, uint32_t sym)
{
    ct->log = 0;
    ct->max_sym = sym;
    ct->state[0] = 0; ct->state[1] = 0;
    ct->sym[sym].delta_nb = 0; ct->sym[sym].delta_find = 0;
}
static uint32_t fse_init_state(const fse_ct *ct, uint32_t sym)
{
    const fse_sym t = ct->sym[sym];
    const uint32_t nb = (t.delta_nb + (1u << 15)) >> 16;
    const uint32_t v = (nb << 16) - t.delta_nb;
    return ct->state[(int)(v >> nb) + t.delta_find];
}
static uint32_t fse_encode(bitw *b, const fse_ct *ct, uint32_t state, uint32_t sym)
{
    const fse_sym t = ct->sym[sym];
    const uint32_t nb = (state + t.delta_nb) >> 16;
    bw_add(b, state, (int)nb);
    return ct->state[(int)(state >> nb) + t.delta_find];
}

static int g_strategy = 1; /* ZSTD_fast = 1, dfast 2, greedy 3, lazy 4, lazy2 5, btlazy2 6, btopt 7, btultra 8, btultra2 9; set by the frame driver */

/* ------------------------------------------------------------ Huffman (compression side) */
typedef struct { uint16_t val; uint8_t nb; } huf_elt;
typedef struct { uint32_t count; uint16_t parent; uint8_t byte; uint8_t nb; } huf_node;

static uint32_t huf_set_max_height(huf_node *node, uint32_t last, uint32_t max_nb)
{
    const uint32_t largest = node[last].nb;
    if (largest <= max_nb) return largest;
    {
        int total = 0, n = (int)last;
        const uint32_t base = 1u << (largest - max_nb);
        while (node[n].nb > max_nb) { total += (int)(base - (1u << (largest - node[n].nb))); node[n].nb = (uint8_t)max_nb; n--; }
        while (node[n].nb == max_nb) n--;
        total >>= (largest - max_nb);
        {
            const uint32_t none = 0xF0F0F0F0u;
            uint32_t rank_last[HUF_LOG_MAX + 2];
            int pos;
            uint32_t cur = max_nb;
            for (pos = 0; pos < HUF_LOG_MAX + 2; pos++) rank_last[pos] = none;
            for (pos = n; pos >= 0; pos--) {
                if (node[pos].nb >= cur) continue;
                cur = node[pos].nb;
                rank_last[max_nb - cur] = (uint32_t)pos;
            }
            while (total > 0) {
                uint32_t dec = (uint32_t)hb((uint32_t)total) + 1;
                for (; dec > 1; dec--) {
                    const uint32_t hp = rank_last[dec], lp = rank_last[dec - 1];
                    if (hp == none) continue;
                    if (lp == none) break;
                    if (node[hp].count <= 2 * node[lp].count) break;
                }
                while (dec <= HUF_LOG_MAX && rank_last[dec] == none) dec++;
                total -= 1 << (dec - 1);
                if (rank_last[dec - 1] == none) rank_last[dec - 1] = rank_last[dec];
                node[rank_last[dec]].nb++;
                if (rank_last[dec] == 0) rank_last[dec] = none;
                else {
                    rank_last[dec]--;
                    if (node[rank_last[dec]].nb != max_nb - dec) rank_last[dec] = none;
                }
            }
            while (total < 0) {
                if (rank_last[1] == none) {
                    while (node[n].nb == max_nb) n--;
                    node[n + 1].nb--;
                    rank_last[1] = (uint32_t)(n + 1);
                    total++;
                    continue;
                }
                node[rank_last[1] + 1].nb--;
                rank_last[1]++;
                total++;
            }
        }
    }
    return max_nb;
}

/* HUF_buildCTable: returns max code length */
static uint32_t huf_build(huf_elt *tree, const uint32_t *count, uint32_t max_sym, uint32_t max_nb)
{
    huf_node node0[2 * 256 + 2];
    huf_node *node = node0 + 1;
    int n, non_null, low_s, low_n, node_nb = 256, node_root;
    memset(node0, 0, sizeof node0);
    /* sort by decreasing count, ties by increasing symbol (bucketed insertion sort) */
    {
        uint32_t base[33], cur[33];
        memset(base, 0, sizeof base);
        for (n = 0; n <= (int)max_sym; n++) base[hb(count[n] + 1)]++;
        for (n = 30; n > 0; n--) base[n - 1] += base[n];
        for (n = 0; n < 32; n++) cur[n] = base[n];
        for (n = 0; n <= (int)max_sym; n++) {
            const uint32_t c = count[n], r = (uint32_t)hb(c + 1) + 1;
            uint32_t pos = cur[r]++;
            while (pos > base[r] && c > node[pos - 1].count) { node[pos] = node[pos - 1]; pos--; }
            node[pos].count = c;
            node[pos].byte = (uint8_t)n;
        }
    }
    non_null = (int)max_sym;
    while (node[non_null].count == 0) non_null--;
    low_s = non_null; node_root = node_nb + low_s - 1; low_n = node_nb;
    node[node_nb].count = node[low_s].count + node[low_s - 1].count;
    node[low_s].parent = node[low_s - 1].parent = (uint16_t)node_nb;
    node_nb++; low_s -= 2;
    for (n = node_nb; n <= node_root; n++) node[n].count = 1u << 30;
    node0[0].count = 1u << 31;
    while (node_nb <= node_root) {
        const int n1 = (node[low_s].count < node[low_n].count) ? low_s-- : low_n++;
        const int n2 = (node[low_s].count < node[low_n].count) ? low_s-- : low_n++;
        node[node_nb].count = node[n1].count + node[n2].count;
        node[n1].parent = node[n2].parent = (uint16_t)node_nb;
        node_nb++;
    }
    node[node_root].nb = 0;
    for (n = node_root - 1; n >= 256; n--) node[n].nb = (uint8_t)(node[node[n].parent].nb + 1);
    for (n = 0; n <= non_null; n++) node[n].nb = (uint8_t)(node[node[n].parent].nb + 1);
    max_nb = huf_set_max_height(node, (uint32_t)non_null, max_nb);
    {
        uint16_t per_rank[HUF_LOG_MAX + 2], val_rank[HUF_LOG_MAX + 2];
        const int alpha = (int)max_sym + 1;
        memset(per_rank, 0, sizeof per_rank);
        memset(val_rank, 0, sizeof val_rank);
        for (n = 0; n <= non_null; n++) per_rank[node[n].nb]++;
        {
            uint16_t min = 0;
            for (n = (int)max_nb; n > 0; n--) { val_rank[n] = min; min = (uint16_t)(min + per_rank[n]); min >>= 1; }
        }
        for (n = 0; n < alpha; n++) tree[node[n].byte].nb = node[n].nb;
        for (n = 0; n < alpha; n++) tree[n].val = val_rank[tree[n].nb]++;
    }
    return max_nb;
}

/* FSE-compress the Huffman weights (HUF_compressWeights); 0 = not compressible, 1 = RLE */
static size_t huf_compress_weights(uint8_t *dst, const uint8_t *w, size_t n)
{
    uint32_t count[HUF_LOG_MAX + 1], max_sym = HUF_LOG_MAX, max_count = 0, s;
    int16_t norm[HUF_LOG_MAX + 1];
    fse_ct ct;
    int log;
    size_t hsz, i;
    bitw b;
    uint32_t s1, s2;
    if (n <= 1) return 0;
    memset(count, 0, sizeof count);
    for (i = 0; i < n; i++) count[w[i]]++;
    while (!count[max_sym]) max_sym--;
    for (s = 0; s <= max_sym; s++) if (count[s] > max_count) max_count = count[s];
    if (max_count == n) return 1;
    if (max_count == 1) return 0;
    log = fse_optimal_log(6, n, max_sym, 2);
    if (fse_normalize(norm, log, count, n, max_sym, 0) <= 0) return 0;
    hsz = fse_write_ncount(dst, norm, max_sym, log);
    if (!hsz) return 0;
    fse_build_ct(&ct, norm, max_sym, log);
    /* FSE_compress_usingCTable: symbols from the end, even index -> state 1, odd -> state 2 */
    if (n <= 2) return 0;
    bw_init(&b, dst + hsz);
    {
        size_t ip = n;
        if (n & 1) {
            s1 = fse_init_state(&ct, w[--ip]);
            s2 = fse_init_state(&ct, w[--ip]);
            s1 = fse_encode(&b, &ct, s1, w[--ip]);
        } else {
            s2 = fse_init_state(&ct, w[--ip]);
            s1 = fse_init_state(&ct, w[--ip]);
        }
        while (ip > 0) {
            s2 = fse_encode(&b, &ct, s2, w[--ip]);
            s1 = fse_encode(&b, &ct, s1, w[--ip]);
        }
        bw_add(&b, s2, ct.log);
        bw_add(&b, s1, ct.log);
    }
    return hsz + bw_close(&b);
}

static size_t huf_write_table(uint8_t *dst, const huf_elt *tree, uint32_t max_sym, uint32_t log)
{
    uint8_t bits2w[HUF_LOG_MAX + 2], w[256];
    uint32_t n;
    size_t hsz;
    bits2w[0] = 0;
    for (n = 1; n < log + 1; n++) bits2w[n] = (uint8_t)(log + 1 - n);
    for (n = 0; n < max_sym; n++) w[n] = bits2w[tree[n].nb];
    hsz = huf_compress_weights(dst + 1, w, max_sym);
    if (hsz > 1 && hsz < max_sym / 2) { dst[0] = (uint8_t)hsz; return hsz + 1; }
    if (max_sym > 128) return 0; /* library returns an error: literals stay uncompressed */
    dst[0] = (uint8_t)(128 + (max_sym - 1));
    w[max_sym] = 0;
    for (n = 0; n < max_sym; n += 2) dst[n / 2 + 1] = (uint8_t)((w[n] << 4) + w[n + 1]);
    return (max_sym + 1) / 2 + 1;
}

static size_t huf_encode_1x(uint8_t *dst, const uint8_t *src, size_t n, const huf_elt *t)
{
    bitw b;
    size_t i;
    bw_init(&b, dst);
    for (i = n; i > 0; i--) bw_add(&b, t[src[i - 1]].val, t[src[i - 1]].nb);
    return bw_close(&b);
}

/* encode with a given table, 1 or 4 streams (HUF_compressCTable_internal); 0 = not compressible */
static size_t huf_encode_streams(uint8_t *dst, size_t hsz, const uint8_t *src, size_t n, const huf_elt *tree, int single)
{
    size_t op = hsz;
    {
        /* the size first: a table of another block (repeat mode) can expand the literals beyond the output buffer, where the
         * library's bit writer stops at the buffer's end and reports "not compressible".  Same verdict here, before any write
         * (found by running the differential hunt under AddressSanitizer). */
        const size_t seg = single ? n : (n + 3) / 4;
        size_t total = hsz + (single ? 0 : 6), k, i;
        for (k = 0; k < (single ? 1u : 4u); k++) {
            const size_t beg = k * seg, end = (single || k == 3) ? n : beg + seg;
            size_t bits = 1; /* end mark */
            for (i = beg; i < end && i < n; i++) bits += tree[src[i]].nb;
            total += (bits + 7) >> 3;
        }
        if (!single && n < 12) return 0;
        if (total >= n - 1) return 0;
    }
    if (single) {
        op += huf_encode_1x(dst + op, src, n, tree);
    } else {
        const size_t seg = (n + 3) / 4;
        size_t k, ip = 0;
        if (n < 12) return 0;
        op += 6;
        for (k = 0; k < 3; k++) {
            const size_t c = huf_encode_1x(dst + op, src + ip, seg, tree);
            dst[hsz + 2 * k] = (uint8_t)c;
            dst[hsz + 2 * k + 1] = (uint8_t)(c >> 8);
            ip += seg; op += c;
        }
        op += huf_encode_1x(dst + op, src + ip, n - ip, tree);
    }
    if (op >= n - 1) return 0;
    return op;
}

/* Huffman table state carried from block to block (ZSTD_hufCTables_t): mode 0 = none,
 * 1 = "check" (a table exists and may be reused if it covers the symbols and is cheaper) */
typedef struct { int mode; huf_elt tree[256]; } huf_state;

/* HUF_compress{1X,4X}_repeat; returns compressed size (incl. table), 0 = not compressible,
 * 1 = RLE; *reused = 1 when the previous table was used (treeless literals) */
static size_t huf_compress(uint8_t *dst, const uint8_t *src, size_t n, int single, huf_state *st, int prefer_repeat,
                           int *reused)
{
    uint32_t count[256], max_sym = 255, largest = 0, s, log;
    huf_elt tree[256];
    size_t i, hsz;
    int mode = st->mode;
    *reused = 0;
    memset(count, 0, sizeof count);
    for (i = 0; i < n; i++) count[src[i]]++;
    while (!count[max_sym]) max_sym--;
    for (s = 0; s <= max_sym; s++) if (count[s] > largest) largest = count[s];
    if (largest == n) { dst[0] = src[0]; return 1; }
    if (largest <= (n >> 7) + 4) return 0;
    if (mode == 1) { /* HUF_validateCTable */
        int bad = 0;
        for (s = 0; s <= max_sym; s++) bad |= (count[s] != 0) & (st->tree[s].nb == 0);
        if (bad) mode = 0;
    }
    if (prefer_repeat && mode != 0) { *reused = 1; return huf_encode_streams(dst, 0, src, n, st->tree, single); }
    log = (uint32_t)fse_optimal_log(HUF_LOG_DEFAULT, n, max_sym, 1);
    memset(tree, 0, sizeof tree);
    log = huf_build(tree, count, max_sym, log);
    hsz = huf_write_table(dst, tree, max_sym, log);
    if (hsz == 0) return 0;
    if (mode != 0) {
        size_t old_bits = 0, new_bits = 0;
        for (s = 0; s <= max_sym; s++) { old_bits += (size_t)st->tree[s].nb * count[s]; new_bits += (size_t)tree[s].nb * count[s]; }
        if ((old_bits >> 3) <= hsz + (new_bits >> 3) || hsz + 12 >= n) {
            *reused = 1;
            return huf_encode_streams(dst, 0, src, n, st->tree, single);
        }
    }
    if (hsz + 12 >= n) return 0;
    st->mode = 0; /* *repeat = HUF_repeat_none: the new table replaces the old one */
    memcpy(st->tree, tree, sizeof tree);
    return huf_encode_streams(dst, hsz, src, n, tree, single);
}

/* ------------------------------------------------------------ literals section */
static size_t min_gain(size_t n) { return (n >> (g_strategy >= 8 ? g_strategy - 1 : 6)) + 2; } /* ZSTD_minGain: btultra and btultra2 accept smaller gains */

static size_t lit_raw(uint8_t *dst, const uint8_t *src, size_t n)
{
    const size_t fl = 1 + (n > 31) + (n > 4095);
    if (fl == 1) dst[0] = (uint8_t)(0 + (n << 3));
    else if (fl == 2) { const uint32_t h = (uint32_t)(0 + (1 << 2) + (n << 4)); dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); }
    else { const uint32_t h = (uint32_t)(0 + (3 << 2) + (n << 4)); dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16); }
    memcpy(dst + fl, src, n);
    return fl + n;
}
static size_t lit_rle(uint8_t *dst, const uint8_t *src, size_t n)
{
    const size_t fl = 1 + (n > 31) + (n > 4095);
    if (fl == 1) dst[0] = (uint8_t)(1 + (n << 3));
    else if (fl == 2) { const uint32_t h = (uint32_t)(1 + (1 << 2) + (n << 4)); dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); }
    else { const uint32_t h = (uint32_t)(1 + (3 << 2) + (n << 4)); dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16); }
    dst[fl] = src[0];
    return fl + 1;
}

/* ZSTD_compressLiterals: `prev` is the confirmed table state, `next` receives the state this
 * block would leave behind (committed by the caller only if the block is emitted compressed) */
static size_t compress_literals(uint8_t *dst, const uint8_t *src, size_t n, const huf_state *prev, huf_state *next,
                                int disable)
{
    const size_t lh = 3 + (n >= 1024) + (n >= 16384);
    const int single = n < 256;
    int reused = 0;
    uint32_t htype = 2;
    size_t c;
    *next = *prev;
    /* ZSTD_disableLiteralsCompression: strategy fast with targetLength > 0 (negative levels) */
    if (disable) return lit_raw(dst, src, n);
    if (n <= 63) return lit_raw(dst, src, n);
    c = huf_compress(dst + lh, src, n, single, next, g_strategy < 4 ? n <= 1024 : 0, &reused);
    if (reused) htype = 3;
    if (c == 0 || c >= n - min_gain(n)) { *next = *prev; return lit_raw(dst, src, n); }
    if (c == 1) { *next = *prev; return lit_rle(dst, src, n); }
    if (htype == 2) next->mode = 1; /* a freshly built table is "check" for the next block */
    if (lh == 3) {
        const uint32_t h = htype + ((uint32_t)(!single) << 2) + ((uint32_t)n << 4) + ((uint32_t)c << 14);
        dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16);
    } else if (lh == 4) {
        const uint32_t h = htype + (2u << 2) + ((uint32_t)n << 4) + ((uint32_t)c << 18);
        dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16); dst[3] = (uint8_t)(h >> 24);
    } else {
        const uint32_t h = htype + (3u << 2) + ((uint32_t)n << 4) + ((uint32_t)c << 22);
        dst[0] = (uint8_t)h; dst[1] = (uint8_t)(h >> 8); dst[2] = (uint8_t)(h >> 16); dst[3] = (uint8_t)(h >> 24);
        dst[4] = (uint8_t)(c >> 10);
    }
    return lh + c;
}

/* ------------------------------------------------------------ sequences */
typedef struct { uint32_t off; uint16_t ll, ml; } seq_t; /* off = offCode + 1, ml = matchLength - 3 */

static const uint8_t LL_bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6,
    7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint8_t ML_bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const int16_t LL_def[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3,
    2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t ML_def[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t OF_def[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1,
    -1, -1, -1};

static uint32_t ll_code(uint32_t ll)
{
    static const uint8_t c[64] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 16, 17, 17, 18, 18, 19,
        19, 20, 20, 20, 20, 21, 21, 21, 21, 22, 22, 22, 22, 22, 22, 22, 22, 23, 23, 23, 23, 23, 23, 23, 23, 24, 24,
        24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24};
    return ll > 63 ? (uint32_t)hb(ll) + 19 : c[ll];
}
static uint32_t ml_code(uint32_t mb)
{
    static const uint8_t c[128] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22,
        23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 32, 33, 33, 34, 34, 35, 35, 36, 36, 36, 36, 37, 37, 37, 37, 38, 38,
        38, 38, 38, 38, 38, 38, 39, 39, 39, 39, 39, 39, 39, 39, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40,
        40, 40, 40, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 42, 42, 42, 42, 42, 42, 42, 42,
        42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42};
    return mb > 127 ? (uint32_t)hb(mb) + 36 : c[mb];
}

enum { SET_BASIC = 0, SET_RLE = 1, SET_COMPRESSED = 2, SET_REPEAT = 3 };

/* ---- ZSTD_selectEncodingType.  Strategies below `lazy`: thresholds; `lazy` and above: estimated costs, and
 * the previous block's table may be repeated (FSE_repeat_check) ---- */
#define COST_ERR ((size_t)-1)
static unsigned inv_prob_log256(unsigned x) /* kInverseProbabilityLog256[x] = (unsigned)(-log2(x / 256.) * 256) */
{
    static unsigned tab[256];
    static int ready = 0;
    if (!ready) {
        unsigned i;
        tab[0] = 0;
        for (i = 1; i < 256; i++) tab[i] = (unsigned)((8.0 - log2((double)i)) * 256.0);
        ready = 1;
    }
    return tab[x];
}
static size_t entropy_cost(const uint32_t *count, uint32_t max, size_t total)
{
    unsigned cost = 0, s;
    for (s = 0; s <= max; s++) {
        unsigned norm = (unsigned)((256 * count[s]) / total);
        if (count[s] != 0 && norm == 0) norm = 1;
        cost += count[s] * inv_prob_log256(norm);
    }
    return cost >> 8;
}
static size_t cross_entropy_cost(const int16_t *norm, unsigned acc_log, const uint32_t *count, uint32_t max)
{
    const unsigned shift = 8 - acc_log;
    size_t cost = 0;
    unsigned s;
    for (s = 0; s <= max; s++) {
        const unsigned nacc = norm[s] != -1 ? (unsigned)norm[s] : 1;
        cost += count[s] * inv_prob_log256(nacc << shift);
    }
    return cost >> 8;
}
static size_t fse_bit_cost(const fse_ct *ct, const uint32_t *count, uint32_t max)
{
    size_t cost = 0;
    unsigned s;
    if (ct->max_sym < max) return COST_ERR;
    for (s = 0; s <= max; s++) {
        const uint32_t tlog = (uint32_t)ct->log, bad = (tlog + 1) << 8;
        const uint32_t min_nb = ct->sym[s].delta_nb >> 16;
        const uint32_t threshold = (min_nb + 1) << 16;
        const uint32_t from_thr = threshold - (ct->sym[s].delta_nb + (1u << tlog));
        const uint32_t norm_from_thr = (from_thr << 8) >> tlog;
        const uint32_t bit_cost = (min_nb + 1) * 256u - norm_from_thr;
        if (count[s] == 0) continue;
        if (bit_cost >= bad) return COST_ERR;
        cost += (size_t)count[s] * bit_cost;
    }
    return cost >> 8;
}
static size_t ncount_cost(const uint32_t *count, uint32_t max, size_t nseq, int fse_log)
{
    uint8_t wksp[512];
    int16_t norm[MaxML + 1];
    const int log = fse_optimal_log(fse_log, nseq, max, 2);
    size_t sz;
    if (fse_normalize(norm, log, count, nseq, max, nseq >= 2048) <= 0) return COST_ERR;
    sz = fse_write_ncount(wksp, norm, max, log);
    return sz ? sz : COST_ERR;
}
/* *rep_mode: 0 none, 1 check (the previous compressed block left a usable table) */
static int select_type(const uint32_t *count, uint32_t max, size_t most, size_t nseq, int fse_log, const fse_ct *prev, int *rep_mode,
                       const int16_t *def_norm, int def_log, int def_allowed)
{
    if (most == nseq) { *rep_mode = 0; return (def_allowed && nseq <= 2) ? SET_BASIC : SET_RLE; }
    if (g_strategy < 4) {
        if (def_allowed) {
            const size_t mult = 10 - (size_t)g_strategy;
            const size_t dyn_min = (((size_t)1 << def_log) * mult) >> 3;
            if (nseq < dyn_min || most < (nseq >> (def_log - 1))) { *rep_mode = 0; return SET_BASIC; }
        }
    } else {
        const size_t basic = def_allowed ? cross_entropy_cost(def_norm, (unsigned)def_log, count, max) : COST_ERR;
        const size_t repeat = *rep_mode != 0 ? fse_bit_cost(prev, count, max) : COST_ERR;
        const size_t nc = ncount_cost(count, max, nseq, fse_log);
        const size_t compressed = (nc << 3) + entropy_cost(count, max, nseq);
        if (basic <= repeat && basic <= compressed) { *rep_mode = 0; return SET_BASIC; }
        if (repeat <= compressed) return SET_REPEAT;
    }
    *rep_mode = 1;
    return SET_COMPRESSED;
}

/* ZSTD_buildCTable; returns bytes written to dst (table description) or (size_t)-1 on error */
static size_t build_ctable(uint8_t *dst, fse_ct *ct, int fse_log, int type, uint32_t *count, uint32_t max,
                           const uint8_t *codes, size_t nseq, const int16_t *def_norm, int def_log, uint32_t def_max)
{
    if (type == SET_REPEAT) return 0; /* *ct already holds the previous block's table */
    if (type == SET_RLE) { fse_build_ct_rle(ct, max); dst[0] = codes[0]; return 1; }
    if (type == SET_BASIC) { fse_build_ct(ct, def_norm, def_max, def_log); return 0; }
    {
        int16_t norm[MaxML + 1];
        size_t n1 = nseq, sz;
        const int log = fse_optimal_log(fse_log, nseq, max, 2);
        if (count[codes[nseq - 1]] > 1) { count[codes[nseq - 1]]--; n1--; }
        if (fse_normalize(norm, log, count, n1, max, n1 >= 2048) <= 0) return (size_t)-1;
        sz = fse_write_ncount(dst, norm, max, log);
        if (!sz) return (size_t)-1;
        fse_build_ct(ct, norm, max, log);
        return sz;
    }
}

/* sequence tables + repeat modes that a compressed block leaves to the next one of its frame */
typedef struct { fse_ct ll, of, ml; int rep_ll, rep_of, rep_ml; } fse_state;

/* literals + sequences of one block -> compressed block body; 0 = emit a raw block */
static size_t compress_sequences(uint8_t *dst, const seq_t *seqs, size_t nseq, const uint8_t *lits, size_t nlit,
                                 size_t src_size, int long_pos, int long_kind, const huf_state *hprev, huf_state *hnext,
                                 const fse_state *fprev, fse_state *fnext, int disable_lit)
{
    static uint8_t llc[ZBLOCK_MAX / 3 + 8], mlc[ZBLOCK_MAX / 3 + 8], ofc[ZBLOCK_MAX / 3 + 8];
#define ct_ll (fnext->ll)
#define ct_of (fnext->of)
#define ct_ml (fnext->ml)
    uint32_t count[MaxML + 1], max, s;
    size_t op, i, most;
    uint8_t *seq_head, *last_ncount = NULL;
    int tll, tof, tml;
    *fnext = *fprev;
    op = compress_literals(dst, lits, nlit, hprev, hnext, disable_lit);
    if (nseq < 128) dst[op++] = (uint8_t)nseq;
    else if (nseq < 0x7F00) { dst[op] = (uint8_t)((nseq >> 8) + 0x80); dst[op + 1] = (uint8_t)nseq; op += 2; }
    else { dst[op] = 0xFF; dst[op + 1] = (uint8_t)(nseq - 0x7F00); dst[op + 2] = (uint8_t)((nseq - 0x7F00) >> 8); op += 3; }
    if (nseq == 0) goto check;
    seq_head = dst + op++;
    for (i = 0; i < nseq; i++) {
        llc[i] = (uint8_t)ll_code(seqs[i].ll);
        ofc[i] = (uint8_t)hb(seqs[i].off);
        mlc[i] = (uint8_t)ml_code(seqs[i].ml);
    }
    if (long_kind == 1) llc[long_pos] = MaxLL;
    if (long_kind == 2) mlc[long_pos] = MaxML;
#define HIST(codes, maxv)                                                                                 \
    do {                                                                                                  \
        memset(count, 0, sizeof count);                                                                   \
        for (i = 0; i < nseq; i++) count[codes[i]]++;                                                     \
        max = (maxv);                                                                                     \
        while (!count[max]) max--;                                                                        \
        most = 0;                                                                                         \
        for (s = 0; s <= max; s++) if (count[s] > most) most = count[s];                                  \
    } while (0)
    {
        size_t sz;
        HIST(llc, MaxLL);
        tll = select_type(count, max, most, nseq, LLFSELog, &fprev->ll, &fnext->rep_ll, LL_def, 6, 1);
        sz = build_ctable(dst + op, &ct_ll, LLFSELog, tll, count, max, llc, nseq, LL_def, 6, MaxLL);
        if (sz == (size_t)-1) return 0;
        if (tll == SET_COMPRESSED) last_ncount = dst + op;
        op += sz;
        HIST(ofc, MaxOff);
        tof = select_type(count, max, most, nseq, OffFSELog, &fprev->of, &fnext->rep_of, OF_def, 5, max <= DefaultMaxOff);
        sz = build_ctable(dst + op, &ct_of, OffFSELog, tof, count, max, ofc, nseq, OF_def, 5, DefaultMaxOff);
        if (sz == (size_t)-1) return 0;
        if (tof == SET_COMPRESSED) last_ncount = dst + op;
        op += sz;
        HIST(mlc, MaxML);
        tml = select_type(count, max, most, nseq, MLFSELog, &fprev->ml, &fnext->rep_ml, ML_def, 6, 1);
        sz = build_ctable(dst + op, &ct_ml, MLFSELog, tml, count, max, mlc, nseq, ML_def, 6, MaxML);
        if (sz == (size_t)-1) return 0;
        if (tml == SET_COMPRESSED) last_ncount = dst + op;
        op += sz;
    }
#undef HIST
    *seq_head = (uint8_t)((tll << 6) + (tof << 4) + (tml << 2));
    {
        bitw b;
        uint32_t sm, so, sl;
        size_t n = nseq - 1, bs;
        bw_init(&b, dst + op);
        b.cap = src_size + 32 > op ? src_size + 32 - op : 0; /* the library's writer stops at the end of its buffer; the size test below gives the same verdict */
        sm = fse_init_state(&ct_ml, mlc[n]);
        so = fse_init_state(&ct_of, ofc[n]);
        sl = fse_init_state(&ct_ll, llc[n]);
        bw_add(&b, seqs[n].ll, LL_bits[llc[n]]);
        bw_add(&b, seqs[n].ml, ML_bits[mlc[n]]);
        bw_add(&b, seqs[n].off, ofc[n]);
        while (n-- > 0) {
            so = fse_encode(&b, &ct_of, so, ofc[n]);
            sm = fse_encode(&b, &ct_ml, sm, mlc[n]);
            sl = fse_encode(&b, &ct_ll, sl, llc[n]);
            bw_add(&b, seqs[n].ll, LL_bits[llc[n]]);
            bw_add(&b, seqs[n].ml, ML_bits[mlc[n]]);
            bw_add(&b, seqs[n].off, ofc[n]);
        }
        bw_add(&b, sm, ct_ml.log);
        bw_add(&b, so, ct_of.log);
        bw_add(&b, sl, ct_ll.log);
        bs = bw_close(&b);
        op += bs;
        if (last_ncount && (size_t)(dst + op - last_ncount) < 4) return 0;
    }
check:
    if (op >= src_size - min_gain(src_size)) return 0;
    return op;
#undef ct_ll
#undef ct_of
#undef ct_ml
}

/* ------------------------------------------------------------ match finder: strategy `fast` */
typedef struct { int wlog, clog, hlog, slog, mml, tlen, dfast, lazy_depth; /* lazy_depth: -1 none, 0 greedy, 1 lazy, 2 lazy2 */
                 int bt;  /* 1: the binary-tree searcher (strategy btlazy2 with the lazy2 parser, and the optimal-parser strategies) */
                 int opt; /* 0, or 1 btopt, 2 btultra, 3 btultra2 */ } cpar;

static uint32_t hash_ptr(const uint8_t *p, int hlog, int mls)
{
    switch (mls) {
    default:
    case 4: return (rd32(p) * 2654435761u) >> (32 - hlog);
    case 5: return (uint32_t)(((rd64(p) << 24) * 889523592379ull) >> (64 - hlog));
    case 6: return (uint32_t)(((rd64(p) << 16) * 227718039650203ull) >> (64 - hlog));
    case 7: return (uint32_t)(((rd64(p) << 8) * 58295818150454627ull) >> (64 - hlog));
    case 8: return (uint32_t)((rd64(p) * 0xCF1BBCDCB7A56463ull) >> (64 - hlog));
    }
}
static size_t count_match(const uint8_t *a, const uint8_t *b, const uint8_t *end)
{
    const uint8_t *s = a;
    while (a < end && *a == *b) { a++; b++; }
    return (size_t)(a - s);
}

typedef struct {
    seq_t *seqs; size_t nseq;
    uint8_t *lits; size_t nlit;
    int long_pos, long_kind;
} seqstore;

static void store_seq(seqstore *ss, size_t ll, const uint8_t *lit, uint32_t offcode, size_t mlbase)
{
    memcpy(ss->lits + ss->nlit, lit, ll);
    ss->nlit += ll;
    if (ll > 0xFFFF) { ss->long_kind = 1; ss->long_pos = (int)ss->nseq; }
    if (mlbase > 0xFFFF) { ss->long_kind = 2; ss->long_pos = (int)ss->nseq; }
    ss->seqs[ss->nseq].ll = (uint16_t)ll;
    ss->seqs[ss->nseq].off = offcode + 1;
    ss->seqs[ss->nseq].ml = (uint16_t)mlbase;
    ss->nseq++;
}

/* base = src - 1 (the library gives the first input byte index 1) */
static size_t block_fast(uint32_t *table, const cpar *cp, const uint8_t *base, const uint8_t *istart, size_t n,
                         uint32_t rep[3], seqstore *ss, uint32_t dict_limit)
{
    const int hlog = cp->hlog, mls = cp->mml < 4 ? 4 : (cp->mml > 7 ? 7 : cp->mml);
    const size_t step_size = (size_t)cp->tlen + !cp->tlen + 1;
    const uint8_t *ip0 = istart, *ip1, *anchor = istart;
    const uint32_t end_index = (uint32_t)(istart - base) + (uint32_t)n;
    const uint32_t max_dist = 1u << cp->wlog;
    const uint32_t prefix_idx = (end_index - dict_limit > max_dist) ? end_index - max_dist : dict_limit;
    const uint8_t *prefix = base + prefix_idx;
    const uint8_t *iend = istart + n, *ilimit = iend - HASH_READ;
    uint32_t off1 = rep[0], off2 = rep[1], saved = 0;

    ip0 += (ip0 == prefix);
    ip1 = ip0 + 1;
    {
        const uint32_t cur = (uint32_t)(ip0 - base);
        const uint32_t wlow = (cur - dict_limit > max_dist) ? cur - max_dist : dict_limit;
        const uint32_t max_rep = cur - wlow;
        if (off2 > max_rep) { saved = off2; off2 = 0; }
        if (off1 > max_rep) { saved = off1; off1 = 0; }
    }
    while (ip1 < ilimit) {
        size_t mlen;
        const uint8_t *ip2 = ip0 + 2;
        const uint32_t h0 = hash_ptr(ip0, hlog, mls), h1 = hash_ptr(ip1, hlog, mls);
        const uint32_t v0 = rd32(ip0), v1 = rd32(ip1);
        const uint32_t cur0 = (uint32_t)(ip0 - base), cur1 = (uint32_t)(ip1 - base);
        const uint32_t mi0 = table[h0], mi1 = table[h1];
        const uint8_t *rep_m = ip2 - off1;
        const uint8_t *m0 = base + mi0, *m1 = base + mi1;
        uint32_t offcode;
        table[h0] = cur0;
        table[h1] = cur1;
        if ((off1 > 0) && rd32(rep_m) == rd32(ip2)) {
            mlen = (ip2[-1] == rep_m[-1]) ? 1 : 0;
            ip0 = ip2 - mlen;
            m0 = rep_m - mlen;
            mlen += 4;
            offcode = 0;
            goto match;
        }
        if (mi0 > prefix_idx && rd32(m0) == v0) goto offset;
        if (mi1 > prefix_idx && rd32(m1) == v1) { ip0 = ip1; m0 = m1; goto offset; }
        {
            const size_t step = ((size_t)(ip0 - anchor) >> (8 - 1)) + step_size;
            ip0 += step;
            ip1 += step;
            continue;
        }
    offset:
        off2 = off1;
        off1 = (uint32_t)(ip0 - m0);
        offcode = off1 + REP_MOVE;
        mlen = 4;
        while ((ip0 > anchor) && (m0 > prefix) && ip0[-1] == m0[-1]) { ip0--; m0--; mlen++; }
    match:
        mlen += count_match(ip0 + mlen, m0 + mlen, iend);
        store_seq(ss, (size_t)(ip0 - anchor), anchor, offcode, mlen - MINMATCH);
        ip0 += mlen;
        anchor = ip0;
        if (ip0 <= ilimit) {
            table[hash_ptr(base + cur0 + 2, hlog, mls)] = cur0 + 2;
            table[hash_ptr(ip0 - 2, hlog, mls)] = (uint32_t)(ip0 - 2 - base);
            if (off2 > 0) {
                while (ip0 <= ilimit && rd32(ip0) == rd32(ip0 - off2)) {
                    const size_t rlen = count_match(ip0 + 4, ip0 + 4 - off2, iend) + 4;
                    const uint32_t t = off2; off2 = off1; off1 = t;
                    table[hash_ptr(ip0, hlog, mls)] = (uint32_t)(ip0 - base);
                    ip0 += rlen;
                    store_seq(ss, 0, anchor, 0, rlen - MINMATCH);
                    anchor = ip0;
                }
            }
        }
        ip1 = ip0 + 1;
    }
    rep[0] = off1 ? off1 : saved;
    rep[1] = off2 ? off2 : saved;
    return (size_t)(iend - anchor);
}

/* ------------------------------------------------------------ match finder: strategy `dfast`
 * (libzstd 1.4.8 ZSTD_compressBlock_doubleFast, no dictionary): a long table hashed on 8 bytes and a short
 * one on mml bytes, both updated at every visited position; order of tests: repeat offset at ip+1, long
 * match at ip, then (only if the short candidate matches 4 bytes) long match at ip+1, else the short one. */
static size_t block_dfast(uint32_t *tlong, uint32_t *tshort, const cpar *cp, const uint8_t *base, const uint8_t *istart,
                          size_t n, uint32_t rep[3], seqstore *ss, uint32_t dict_limit)
{
    const int hl = cp->hlog, hs = cp->clog, mls = cp->mml < 4 ? 4 : (cp->mml > 7 ? 7 : cp->mml);
    const uint8_t *ip = istart, *anchor = istart;
    const uint32_t end_index = (uint32_t)(istart - base) + (uint32_t)n;
    const uint32_t max_dist = 1u << cp->wlog;
    const uint32_t prefix_idx = (end_index - dict_limit > max_dist) ? end_index - max_dist : dict_limit;
    const uint8_t *prefix = base + prefix_idx;
    const uint8_t *iend = istart + n, *ilimit = iend - HASH_READ;
    uint32_t off1 = rep[0], off2 = rep[1], saved = 0;

    ip += (ip == prefix);
    {
        const uint32_t cur = (uint32_t)(ip - base);
        const uint32_t wlow = (cur - dict_limit > max_dist) ? cur - max_dist : dict_limit;
        const uint32_t max_rep = cur - wlow;
        if (off2 > max_rep) { saved = off2; off2 = 0; }
        if (off1 > max_rep) { saved = off1; off1 = 0; }
    }
    while (ip < ilimit) {
        size_t mlen;
        uint32_t offset;
        const uint32_t h2 = hash_ptr(ip, hl, 8), h = hash_ptr(ip, hs, mls);
        const uint32_t cur = (uint32_t)(ip - base);
        const uint32_t mil = tlong[h2], mis = tshort[h];
        const uint8_t *ml = base + mil, *m = base + mis;
        tlong[h2] = tshort[h] = cur;
        if (off1 > 0 && rd32(ip + 1 - off1) == rd32(ip + 1)) {
            mlen = count_match(ip + 1 + 4, ip + 1 + 4 - off1, iend) + 4;
            ip++;
            store_seq(ss, (size_t)(ip - anchor), anchor, 0, mlen - MINMATCH);
            goto stored;
        }
        if (mil > prefix_idx && rd64(ml) == rd64(ip)) {
            mlen = count_match(ip + 8, ml + 8, iend) + 8;
            offset = (uint32_t)(ip - ml);
            while (ip > anchor && ml > prefix && ip[-1] == ml[-1]) { ip--; ml--; mlen++; }
            goto found;
        }
        if (mis > prefix_idx && rd32(m) == rd32(ip)) {
            const uint32_t hl3 = hash_ptr(ip + 1, hl, 8);
            const uint32_t mil3 = tlong[hl3];
            const uint8_t *ml3 = base + mil3;
            tlong[hl3] = cur + 1;
            if (mil3 > prefix_idx && rd64(ml3) == rd64(ip + 1)) {
                mlen = count_match(ip + 9, ml3 + 8, iend) + 8;
                ip++;
                offset = (uint32_t)(ip - ml3);
                while (ip > anchor && ml3 > prefix && ip[-1] == ml3[-1]) { ip--; ml3--; mlen++; }
                goto found;
            }
            mlen = count_match(ip + 4, m + 4, iend) + 4;
            offset = (uint32_t)(ip - m);
            while (ip > anchor && m > prefix && ip[-1] == m[-1]) { ip--; m--; mlen++; }
            goto found;
        }
        ip += ((size_t)(ip - anchor) >> 8) + 1;
        continue;
    found:
        off2 = off1;
        off1 = offset;
        store_seq(ss, (size_t)(ip - anchor), anchor, offset + REP_MOVE, mlen - MINMATCH);
    stored:
        ip += mlen;
        anchor = ip;
        if (ip <= ilimit) {
            const uint32_t ins = cur + 2;
            tlong[hash_ptr(base + ins, hl, 8)] = ins;
            tlong[hash_ptr(ip - 2, hl, 8)] = (uint32_t)(ip - 2 - base);
            tshort[hash_ptr(base + ins, hs, mls)] = ins;
            tshort[hash_ptr(ip - 1, hs, mls)] = (uint32_t)(ip - 1 - base);
            while (ip <= ilimit && off2 > 0 && rd32(ip) == rd32(ip - off2)) {
                const size_t rlen = count_match(ip + 4, ip + 4 - off2, iend) + 4;
                const uint32_t t = off2; off2 = off1; off1 = t;
                tshort[hash_ptr(ip, hs, mls)] = (uint32_t)(ip - base);
                tlong[hash_ptr(ip, hl, 8)] = (uint32_t)(ip - base);
                store_seq(ss, 0, anchor, 0, rlen - MINMATCH);
                ip += rlen;
                anchor = ip;
            }
        }
    }
    rep[0] = off1 ? off1 : saved;
    rep[1] = off2 ? off2 : saved;
    return (size_t)(iend - anchor);
}

/* ------------------------------------------------------------ match finder: strategies `greedy`, `lazy`, `lazy2`
 * (libzstd 1.4.8 ZSTD_compressBlock_lazy_generic over the hash-chain searcher ZSTD_HcFindBestMatch, no
 * dictionary): every position up to the one searched is inserted into a hash table + chain table; a search
 * walks at most 2^searchLog chain links and keeps the longest match; depth 1/2 retry at ip+1 (ip+2) and keep
 * the candidate whose gain estimate is better. */
typedef struct { uint32_t *hash, *chain; uint32_t next_to_update;
                 uint32_t *hash3; int hlog3; uint32_t low; /* the optimal-parser strategies: 3-byte hash table, lowest valid index */ } hc_state;

static uint32_t hc_insert_find(hc_state *hc, const cpar *cp, const uint8_t *base, uint32_t target, int mls)
{
    const uint32_t cmask = (1u << cp->clog) - 1u;
    uint32_t idx = hc->next_to_update;
    while (idx < target) {
        const uint32_t h = hash_ptr(base + idx, cp->hlog, mls);
        hc->chain[idx & cmask] = hc->hash[h];
        hc->hash[h] = idx;
        idx++;
    }
    hc->next_to_update = target;
    return hc->hash[hash_ptr(base + target, cp->hlog, mls)];
}

static size_t hc_find_best(hc_state *hc, const cpar *cp, const uint8_t *base, const uint8_t *ip, const uint8_t *ilimit_end,
                           size_t *offset_ptr, int mls)
{
    const uint32_t csize = 1u << cp->clog, cmask = csize - 1u;
    const uint32_t cur = (uint32_t)(ip - base);
    const uint32_t max_dist = 1u << cp->wlog;
    const uint32_t lowest_valid = 1; /* window.lowLimit */
    const uint32_t low_limit = (cur - lowest_valid > max_dist) ? cur - max_dist : lowest_valid;
    const uint32_t min_chain = cur > csize ? cur - csize : 0;
    uint32_t attempts = 1u << cp->slog;
    size_t ml = 4 - 1;
    uint32_t mi = hc_insert_find(hc, cp, base, cur, mls);
    for (; (mi >= low_limit) && (attempts > 0); attempts--) {
        size_t cml = 0;
        const uint8_t *match = base + mi;
        if (match[ml] == ip[ml]) cml = count_match(ip, match, ilimit_end);
        if (cml > ml) {
            ml = cml;
            *offset_ptr = cur - mi + REP_MOVE;
            if (ip + cml == ilimit_end) break;
        }
        if (mi <= min_chain) break;
        mi = hc->chain[mi & cmask];
    }
    return ml;
}

/* ------------------------------------------------------------ match finder: strategy `btlazy2` (libzstd 1.4.8 zstd_lazy.c:
 * ZSTD_updateDUBT, ZSTD_insertDUBT1, ZSTD_DUBT_findBestMatch, ZSTD_BtFindBestMatch; no dictionary).  The chain table is a
 * binary tree of 2^(chainLog-1) nodes, two links per position (smaller / larger suffixes); positions are first chained
 * UNSORTED (second link = the mark 1) and sorted into the tree in batches when a search runs into them. */
#define DUBT_UNSORTED_MARK 1u

static void dubt_update(hc_state *hc, const cpar *cp, const uint8_t *base, uint32_t target, int mls)
{
    uint32_t *const bt = hc->chain;
    const uint32_t bt_mask = (1u << (cp->clog - 1)) - 1u;
    uint32_t idx = hc->next_to_update;
    for (; idx < target; idx++) {
        const uint32_t h = hash_ptr(base + idx, cp->hlog, mls);
        const uint32_t mi = hc->hash[h];
        uint32_t *const next_cand = bt + 2u * (idx & bt_mask);
        hc->hash[h] = idx;
        next_cand[0] = mi;                /* the tree slot used like a chain link */
        next_cand[1] = DUBT_UNSORTED_MARK;
    }
    hc->next_to_update = target;
}

/* sort one position that is chained but not yet in the tree */
static void dubt_insert1(hc_state *hc, const cpar *cp, const uint8_t *base, uint32_t cur, const uint8_t *iend, uint32_t nb_compares,
                         uint32_t bt_low)
{
    uint32_t *const bt = hc->chain;
    const uint32_t bt_mask = (1u << (cp->clog - 1)) - 1u;
    size_t common_smaller = 0, common_larger = 0;
    const uint8_t *const ip = base + cur;
    uint32_t *smaller_ptr = bt + 2u * (cur & bt_mask);
    uint32_t *larger_ptr = smaller_ptr + 1;
    uint32_t mi = *smaller_ptr; /* the next sorted candidate; *larger_ptr held the previous unsorted one (saved by the caller) */
    uint32_t dummy;
    const uint32_t window_valid = 1, max_dist = 1u << cp->wlog;
    const uint32_t window_low = (cur - window_valid > max_dist) ? cur - max_dist : window_valid;
    while (nb_compares-- && mi > window_low) {
        uint32_t *const next_ptr = bt + 2u * (mi & bt_mask);
        size_t ml = common_smaller < common_larger ? common_smaller : common_larger;
        const uint8_t *const match = base + mi;
        ml += count_match(ip + ml, match + ml, iend);
        if (ip + ml == iend) break; /* equal: no way to know if smaller or larger: dropped */
        if (match[ml] < ip[ml]) {
            *smaller_ptr = mi;
            common_smaller = ml;
            if (mi <= bt_low) { smaller_ptr = &dummy; break; }
            smaller_ptr = next_ptr + 1;
            mi = next_ptr[1];
        } else {
            *larger_ptr = mi;
            common_larger = ml;
            if (mi <= bt_low) { larger_ptr = &dummy; break; }
            larger_ptr = next_ptr;
            mi = next_ptr[0];
        }
    }
    *smaller_ptr = *larger_ptr = 0;
}

static size_t bt_find_best(hc_state *hc, const cpar *cp, const uint8_t *base, const uint8_t *ip, const uint8_t *iend,
                           size_t *offset_ptr, int mls)
{
    uint32_t *const bt = hc->chain;
    const uint32_t bt_mask = (1u << (cp->clog - 1)) - 1u;
    const uint32_t cur = (uint32_t)(ip - base);
    uint32_t h, mi, window_low, bt_low, unsort_limit, nb_compares, nb_candidates, previous = 0;
    uint32_t *next_cand, *unsorted_mark;
    if (cur < hc->next_to_update) return 0; /* skipped area */
    dubt_update(hc, cp, base, cur, mls);
    h = hash_ptr(ip, cp->hlog, mls);
    mi = hc->hash[h];
    {
        const uint32_t max_dist = 1u << cp->wlog, lowest_valid = 1;
        window_low = (cur - lowest_valid > max_dist) ? cur - max_dist : lowest_valid;
    }
    bt_low = (bt_mask >= cur) ? 0 : cur - bt_mask;
    unsort_limit = bt_low > window_low ? bt_low : window_low;
    next_cand = bt + 2u * (mi & bt_mask);
    unsorted_mark = next_cand + 1;
    nb_compares = 1u << cp->slog;
    nb_candidates = nb_compares;
    /* reach the end of the unsorted candidates (their marks become a reversed chain to come back by) */
    while (mi > unsort_limit && *unsorted_mark == DUBT_UNSORTED_MARK && nb_candidates > 1) {
        *unsorted_mark = previous;
        previous = mi;
        mi = *next_cand;
        next_cand = bt + 2u * (mi & bt_mask);
        unsorted_mark = next_cand + 1;
        nb_candidates--;
    }
    /* the last candidate, if still unsorted, is dropped */
    if (mi > unsort_limit && *unsorted_mark == DUBT_UNSORTED_MARK) *next_cand = *unsorted_mark = 0;
    /* batch sort of the stacked candidates */
    mi = previous;
    while (mi) {
        uint32_t *const next_idx_ptr = bt + 2u * (mi & bt_mask) + 1;
        const uint32_t next_idx = *next_idx_ptr;
        dubt_insert1(hc, cp, base, mi, iend, nb_candidates, unsort_limit);
        mi = next_idx;
        nb_candidates++;
    }
    /* the longest match, inserting the current position on the way */
    {
        size_t common_smaller = 0, common_larger = 0, best = 0;
        uint32_t *smaller_ptr = bt + 2u * (cur & bt_mask);
        uint32_t *larger_ptr = smaller_ptr + 1;
        uint32_t match_end_idx = cur + 8 + 1;
        uint32_t dummy;
        mi = hc->hash[h];
        hc->hash[h] = cur;
        while (nb_compares-- && mi > window_low) {
            uint32_t *const next_ptr = bt + 2u * (mi & bt_mask);
            size_t ml = common_smaller < common_larger ? common_smaller : common_larger;
            const uint8_t *const match = base + mi;
            ml += count_match(ip + ml, match + ml, iend);
            if (ml > best) {
                if (ml > match_end_idx - mi) match_end_idx = mi + (uint32_t)ml;
                if ((4 * (int)(ml - best)) > (int)(hb(cur - mi + 1) - hb((uint32_t)offset_ptr[0] + 1))) {
                    best = ml;
                    *offset_ptr = REP_MOVE + cur - mi;
                }
                if (ip + ml == iend) break; /* equal: dropped, to keep the tree consistent */
            }
            if (match[ml] < ip[ml]) {
                *smaller_ptr = mi;
                common_smaller = ml;
                if (mi <= bt_low) { smaller_ptr = &dummy; break; }
                smaller_ptr = next_ptr + 1;
                mi = next_ptr[1];
            } else {
                *larger_ptr = mi;
                common_larger = ml;
                if (mi <= bt_low) { larger_ptr = &dummy; break; }
                larger_ptr = next_ptr;
                mi = next_ptr[0];
            }
        }
        *smaller_ptr = *larger_ptr = 0;
        hc->next_to_update = match_end_idx - 8; /* skip repetitive patterns */
        return best;
    }
}

static int hb32(uint32_t v) { return hb(v); }

static size_t block_lazy(hc_state *hc, const cpar *cp, const uint8_t *base, const uint8_t *istart, size_t n, uint32_t rep[3],
                         seqstore *ss)
{
    const int depth = cp->lazy_depth;
    const int mls = cp->mml < 4 ? 4 : (cp->mml > 6 ? 6 : cp->mml);
    const uint8_t *ip = istart, *anchor = istart;
    const uint8_t *iend = istart + n, *ilimit = iend - 8;
    const uint8_t *prefix_lowest = base + 1; /* window.dictLimit: not window-limited here */
    uint32_t off1 = rep[0], off2 = rep[1], saved = 0;
    ip += (ip == prefix_lowest);
    {
        const uint32_t cur = (uint32_t)(ip - base);
        const uint32_t max_dist = 1u << cp->wlog;
        const uint32_t wlow = (cur - 1u > max_dist) ? cur - max_dist : 1u;
        const uint32_t max_rep = cur - wlow;
        if (off2 > max_rep) { saved = off2; off2 = 0; }
        if (off1 > max_rep) { saved = off1; off1 = 0; }
    }
    while (ip < ilimit) {
        size_t mlen = 0, offset = 0;
        const uint8_t *start = ip + 1;
        if (off1 > 0 && rd32(ip + 1 - off1) == rd32(ip + 1)) {
            mlen = count_match(ip + 1 + 4, ip + 1 + 4 - off1, iend) + 4;
            if (depth == 0) goto store;
        }
        {
            size_t off_found = 999999999;
            const size_t ml2 = cp->bt ? bt_find_best(hc, cp, base, ip, iend, &off_found, mls) : hc_find_best(hc, cp, base, ip, iend, &off_found, mls);
            if (ml2 > mlen) { mlen = ml2; start = ip; offset = off_found; }
        }
        if (mlen < 4) { ip += ((size_t)(ip - anchor) >> 8) + 1; continue; }
        if (depth >= 1)
            while (ip < ilimit) {
                ip++;
                if (offset && off1 > 0 && rd32(ip) == rd32(ip - off1)) {
                    const size_t ml_rep = count_match(ip + 4, ip + 4 - off1, iend) + 4;
                    const int gain2 = (int)(ml_rep * 3);
                    const int gain1 = (int)(mlen * 3 - (size_t)hb32((uint32_t)offset + 1) + 1);
                    if (ml_rep >= 4 && gain2 > gain1) { mlen = ml_rep; offset = 0; start = ip; }
                }
                {
                    size_t off2f = 999999999;
                    const size_t ml2 = cp->bt ? bt_find_best(hc, cp, base, ip, iend, &off2f, mls) : hc_find_best(hc, cp, base, ip, iend, &off2f, mls);
                    const int gain2 = (int)(ml2 * 4 - (size_t)hb32((uint32_t)off2f + 1));
                    const int gain1 = (int)(mlen * 4 - (size_t)hb32((uint32_t)offset + 1) + 4);
                    if (ml2 >= 4 && gain2 > gain1) { mlen = ml2; offset = off2f; start = ip; continue; }
                }
                if (depth == 2 && ip < ilimit) {
                    ip++;
                    if (offset && off1 > 0 && rd32(ip) == rd32(ip - off1)) {
                        const size_t ml_rep = count_match(ip + 4, ip + 4 - off1, iend) + 4;
                        const int gain2 = (int)(ml_rep * 4);
                        const int gain1 = (int)(mlen * 4 - (size_t)hb32((uint32_t)offset + 1) + 1);
                        if (ml_rep >= 4 && gain2 > gain1) { mlen = ml_rep; offset = 0; start = ip; }
                    }
                    {
                        size_t off2f = 999999999;
                        const size_t ml2 = cp->bt ? bt_find_best(hc, cp, base, ip, iend, &off2f, mls) : hc_find_best(hc, cp, base, ip, iend, &off2f, mls);
                        const int gain2 = (int)(ml2 * 4 - (size_t)hb32((uint32_t)off2f + 1));
                        const int gain1 = (int)(mlen * 4 - (size_t)hb32((uint32_t)offset + 1) + 7);
                        if (ml2 >= 4 && gain2 > gain1) { mlen = ml2; offset = off2f; start = ip; continue; }
                    }
                }
                break;
            }
        if (offset) {
            while (start > anchor && start - (offset - REP_MOVE) > prefix_lowest && start[-1] == (start - (offset - REP_MOVE))[-1]) { start--; mlen++; }
            off2 = off1;
            off1 = (uint32_t)(offset - REP_MOVE);
        }
    store:
        store_seq(ss, (size_t)(start - anchor), anchor, (uint32_t)offset, mlen - MINMATCH);
        anchor = ip = start + mlen;
        while (ip <= ilimit && off2 > 0 && rd32(ip) == rd32(ip - off2)) {
            mlen = count_match(ip + 4, ip + 4 - off2, iend) + 4;
            offset = off2; off2 = off1; off1 = (uint32_t)offset;
            store_seq(ss, 0, anchor, 0, mlen - MINMATCH);
            ip += mlen;
            anchor = ip;
        }
    }
    rep[0] = off1 ? off1 : saved;
    rep[1] = off2 ? off2 : saved;
    return (size_t)(iend - anchor);
}

/* ------------------------------------------------------------ strategies btopt, btultra, btultra2
 * (libzstd 1.4.8 lib/compress/zstd_opt.c: ZSTD_compressBlock_opt_generic and its helpers.)  The binary tree is filled in order
 * (ZSTD_insertBt1), every search returns the whole ladder of matches of increasing length (ZSTD_insertBtAndGetAllMatches), and a
 * forward pass over up to 4096 positions prices every reachable position with running symbol statistics; the cheapest path is
 * then emitted backwards.  optlevel 0: btopt (integer log2 prices, two shortcuts); 2: btultra (fractional prices). */
#define OPT_NUM 4096u
#define OPT_MAX_PRICE (1 << 30)
#define BITCOST 256u
typedef struct { int price; uint32_t off, mlen, litlen, rep[3]; } opt_t;
typedef struct { uint32_t off, len; } match_t;
typedef struct {
    uint32_t lit[256], ll[MaxLL + 1], ml[MaxML + 1], of[MaxOff + 1];
    uint32_t lit_sum, ll_sum, ml_sum, of_sum;
    uint32_t lit_base, ll_base, ml_base, of_base;
    int predef;
} opt_stats;

static uint32_t opt_weight(uint32_t stat, int lvl)
{
    const uint32_t s = stat + 1;
    const int h = hb(s);
    return lvl ? (uint32_t)h * BITCOST + ((s << 8) >> h) : (uint32_t)h * BITCOST;
}
static void opt_set_base(opt_stats *o, int lvl)
{
    o->lit_base = opt_weight(o->lit_sum, lvl);
    o->ll_base = opt_weight(o->ll_sum, lvl);
    o->ml_base = opt_weight(o->ml_sum, lvl);
    o->of_base = opt_weight(o->of_sum, lvl);
}
static uint32_t opt_downscale(uint32_t *t, uint32_t last, int malus)
{
    uint32_t s, sum = 0;
    for (s = 0; s <= last; s++) { t[s] = 1 + (t[s] >> (4 + malus)); sum += t[s]; }
    return sum;
}
static uint32_t opt_upscale(uint32_t *t, uint32_t last)
{
    uint32_t s, sum = 0;
    for (s = 0; s <= last; s++) { t[s] <<= 4; t[s]--; sum += t[s]; }
    return sum;
}
/* ZSTD_rescaleFreqs: first block of a frame: literal counts of the raw block, flat counts for the codes; later blocks:
 * the previous statistics scaled down */
static void opt_rescale(opt_stats *o, const uint8_t *src, size_t n, int lvl)
{
    o->predef = 0;
    if (o->ll_sum == 0) {
        uint32_t k;
        size_t i;
        if (n <= 1024) o->predef = 1;
        memset(o->lit, 0, sizeof o->lit);
        for (i = 0; i < n; i++) o->lit[src[i]]++;
        o->lit_sum = opt_downscale(o->lit, 255, 1);
        for (k = 0; k <= MaxLL; k++) o->ll[k] = 1;
        o->ll_sum = MaxLL + 1;
        for (k = 0; k <= MaxML; k++) o->ml[k] = 1;
        o->ml_sum = MaxML + 1;
        for (k = 0; k <= MaxOff; k++) o->of[k] = 1;
        o->of_sum = MaxOff + 1;
    } else {
        o->lit_sum = opt_downscale(o->lit, 255, 1);
        o->ll_sum = opt_downscale(o->ll, MaxLL, 0);
        o->ml_sum = opt_downscale(o->ml, MaxML, 0);
        o->of_sum = opt_downscale(o->of, MaxOff, 0);
    }
    opt_set_base(o, lvl);
}
static uint32_t opt_lit_cost1(const opt_stats *o, uint8_t c, int lvl)
{
    if (o->predef) return 6 * BITCOST;
    return o->lit_base - opt_weight(o->lit[c], lvl);
}
static uint32_t opt_ll_price(const opt_stats *o, uint32_t ll, int lvl)
{
    if (o->predef) return opt_weight(ll, lvl);
    {
        const uint32_t c = ll_code(ll);
        return LL_bits[c] * BITCOST + o->ll_base - opt_weight(o->ll[c], lvl);
    }
}
static uint32_t opt_match_price(const opt_stats *o, uint32_t off, uint32_t mlen, int lvl)
{
    const uint32_t oc = (uint32_t)hb(off + 1);
    const uint32_t mb = mlen - MINMATCH;
    uint32_t price;
    if (o->predef) return opt_weight(mb, lvl) + (16 + oc) * BITCOST;
    price = oc * BITCOST + (o->of_base - opt_weight(o->of[oc], lvl));
    if (lvl < 2 && oc >= 20) price += (oc - 19) * 2 * BITCOST; /* long offsets handicapped below btultra */
    {
        const uint32_t mc = ml_code(mb);
        price += ML_bits[mc] * BITCOST + (o->ml_base - opt_weight(o->ml[mc], lvl));
    }
    return price + BITCOST / 5;
}
static void opt_update_stats(opt_stats *o, uint32_t ll, const uint8_t *lit, uint32_t offcode, uint32_t mlen)
{
    uint32_t u;
    for (u = 0; u < ll; u++) o->lit[lit[u]] += 2;
    o->lit_sum += ll * 2;
    o->ll[ll_code(ll)]++; o->ll_sum++;
    o->of[hb(offcode + 1)]++; o->of_sum++;
    o->ml[ml_code(mlen - MINMATCH)]++; o->ml_sum++;
}

/* ZSTD_insertBt1: one position into the sorted tree; returns how many positions to advance */
static uint32_t bt_insert1(hc_state *hc, const cpar *cp, const uint8_t *base, uint32_t cur, const uint8_t *iend, int mls)
{
    uint32_t *const bt = hc->chain;
    const uint32_t bt_mask = (1u << (cp->clog - 1)) - 1u;
    const uint8_t *const ip = base + cur;
    const uint32_t h = hash_ptr(ip, cp->hlog, mls);
    uint32_t mi = hc->hash[h];
    size_t common_smaller = 0, common_larger = 0;
    const uint32_t bt_low = bt_mask >= cur ? 0 : cur - bt_mask;
    uint32_t *smaller_ptr = bt + 2u * (cur & bt_mask);
    uint32_t *larger_ptr = smaller_ptr + 1;
    uint32_t dummy;
    uint32_t match_end = cur + 8 + 1;
    size_t best = 8;
    uint32_t nb = 1u << cp->slog;
    hc->hash[h] = cur;
    while (nb-- && mi >= hc->low) {
        uint32_t *const next_ptr = bt + 2u * (mi & bt_mask);
        size_t ml = common_smaller < common_larger ? common_smaller : common_larger;
        const uint8_t *const match = base + mi;
        ml += count_match(ip + ml, match + ml, iend);
        if (ml > best) {
            best = ml;
            if (ml > match_end - mi) match_end = mi + (uint32_t)ml;
        }
        if (ip + ml == iend) break;
        if (match[ml] < ip[ml]) {
            *smaller_ptr = mi;
            common_smaller = ml;
            if (mi <= bt_low) { smaller_ptr = &dummy; break; }
            smaller_ptr = next_ptr + 1;
            mi = next_ptr[1];
        } else {
            *larger_ptr = mi;
            common_larger = ml;
            if (mi <= bt_low) { larger_ptr = &dummy; break; }
            larger_ptr = next_ptr;
            mi = next_ptr[0];
        }
    }
    *smaller_ptr = *larger_ptr = 0;
    {
        uint32_t positions = 0;
        if (best > 384) positions = (uint32_t)(best - 384) < 192u ? (uint32_t)(best - 384) : 192u;
        return positions > match_end - (cur + 8) ? positions : match_end - (cur + 8);
    }
}
static void bt_update_tree(hc_state *hc, const cpar *cp, const uint8_t *base, uint32_t target, const uint8_t *iend, int mls)
{
    uint32_t idx = hc->next_to_update;
    while (idx < target) idx += bt_insert1(hc, cp, base, idx, iend, mls);
    hc->next_to_update = target;
}
static uint32_t hash3_ptr(const uint8_t *p, int hlog) { return ((rd32(p) << 8) * 506832829u) >> (32 - hlog); }

/* ZSTD_BtGetAllMatches + ZSTD_insertBtAndGetAllMatches: the repeat offsets, the 3-byte hash (minMatch 3), then the tree search
 * that inserts the position; matches come out by increasing length */
static uint32_t bt_get_all_matches(match_t *matches, hc_state *hc, const cpar *cp, const uint8_t *base, uint32_t *next3,
                                   const uint8_t *ip, const uint8_t *iend, const uint32_t rep[3], uint32_t ll0,
                                   uint32_t length_to_beat)
{
    const int mls = cp->mml;
    const uint32_t cur = (uint32_t)(ip - base);
    const uint32_t min_match = (mls == 3) ? 3 : 4;
    const uint32_t sufficient = (uint32_t)cp->tlen < OPT_NUM - 1 ? (uint32_t)cp->tlen : OPT_NUM - 1;
    uint32_t *const bt = hc->chain;
    const uint32_t bt_mask = (1u << (cp->clog - 1)) - 1u;
    uint32_t mnum = 0;
    size_t best = length_to_beat - 1;
    if (cur < hc->next_to_update) return 0; /* skipped area */
    bt_update_tree(hc, cp, base, cur, iend, mls);
    {
        const uint32_t h = hash_ptr(ip, cp->hlog, mls);
        uint32_t mi = hc->hash[h];
        size_t common_smaller = 0, common_larger = 0;
        const uint32_t bt_low = bt_mask >= cur ? 0 : cur - bt_mask;
        const uint32_t max_dist = 1u << cp->wlog;
        const uint32_t window_low = (cur - hc->low > max_dist) ? cur - max_dist : hc->low;
        const uint32_t match_low = window_low ? window_low : 1;
        uint32_t *smaller_ptr = bt + 2u * (cur & bt_mask);
        uint32_t *larger_ptr = smaller_ptr + 1;
        uint32_t match_end = cur + 8 + 1;
        uint32_t dummy;
        uint32_t nb = 1u << cp->slog;
        uint32_t rc;
        /* repeat offsets */
        for (rc = ll0; rc < 3 + ll0; rc++) {
            const uint32_t roff = (rc == 3) ? rep[0] - 1 : rep[rc];
            const uint32_t rindex = cur - roff;
            uint32_t rlen = 0;
            if (roff - 1 < cur - hc->low) { /* discards 0 and anything reaching below the prefix start */
                const uint32_t a = min_match == 3 ? rd32(ip) << 8 : rd32(ip);
                const uint32_t b = min_match == 3 ? rd32(ip - roff) << 8 : rd32(ip - roff);
                if (rindex >= window_low && a == b)
                    rlen = (uint32_t)count_match(ip + min_match, ip + min_match - roff, iend) + min_match;
            }
            if (rlen > best) {
                best = rlen;
                matches[mnum].off = rc - ll0;
                matches[mnum].len = rlen;
                mnum++;
                if (rlen > sufficient || ip + rlen == iend) return mnum;
            }
        }
        /* 3-byte matches through their own hash table */
        if (mls == 3 && best < 3) {
            const uint32_t h3 = hash3_ptr(ip, hc->hlog3);
            uint32_t idx = *next3, m3;
            while (idx < cur) { hc->hash3[hash3_ptr(base + idx, hc->hlog3)] = idx; idx++; }
            *next3 = cur;
            m3 = hc->hash3[h3];
            if (m3 >= match_low && cur - m3 < (1u << 18)) {
                const size_t ml = count_match(ip, base + m3, iend);
                if (ml >= 3) {
                    best = ml;
                    matches[0].off = (cur - m3) + REP_MOVE;
                    matches[0].len = (uint32_t)ml;
                    mnum = 1;
                    if (ml > sufficient || ip + ml == iend) { hc->next_to_update = cur + 1; return 1; }
                }
            }
        }
        hc->hash[h] = cur;
        while (nb-- && mi >= match_low) {
            uint32_t *const next_ptr = bt + 2u * (mi & bt_mask);
            size_t ml = common_smaller < common_larger ? common_smaller : common_larger;
            const uint8_t *const match = base + mi;
            ml += count_match(ip + ml, match + ml, iend);
            if (ml > best) {
                if (ml > match_end - mi) match_end = mi + (uint32_t)ml;
                best = ml;
                matches[mnum].off = (cur - mi) + REP_MOVE;
                matches[mnum].len = (uint32_t)ml;
                mnum++;
                if (ml > OPT_NUM || ip + ml == iend) break;
            }
            if (match[ml] < ip[ml]) {
                *smaller_ptr = mi;
                common_smaller = ml;
                if (mi <= bt_low) { smaller_ptr = &dummy; break; }
                smaller_ptr = next_ptr + 1;
                mi = next_ptr[1];
            } else {
                *larger_ptr = mi;
                common_larger = ml;
                if (mi <= bt_low) { larger_ptr = &dummy; break; }
                larger_ptr = next_ptr;
                mi = next_ptr[0];
            }
        }
        *smaller_ptr = *larger_ptr = 0;
        hc->next_to_update = match_end - 8;
    }
    return mnum;
}

static void opt_update_rep(uint32_t out[3], const uint32_t rep[3], uint32_t off, uint32_t ll0)
{
    if (off >= 3) { out[2] = rep[1]; out[1] = rep[0]; out[0] = off - REP_MOVE; }
    else {
        const uint32_t rc = off + ll0;
        if (rc > 0) {
            const uint32_t cur_off = (rc == 3) ? rep[0] - 1 : rep[rc];
            const uint32_t r1 = rep[1], r0 = rep[0], r2 = rep[2];
            out[2] = (rc >= 2) ? r1 : r2;
            out[1] = r0;
            out[0] = cur_off;
        } else { out[0] = rep[0]; out[1] = rep[1]; out[2] = rep[2]; }
    }
}

static size_t block_opt(hc_state *hc, const cpar *cp, opt_stats *o, const uint8_t *base, const uint8_t *istart, size_t n,
                        uint32_t rep[3], seqstore *ss, int lvl)
{
    static opt_t opt[OPT_NUM + 2];
    static match_t matches[OPT_NUM + 2];
    const uint8_t *ip = istart, *anchor = istart;
    const uint8_t *const iend = istart + n, *const ilimit = iend - 8;
    const uint32_t sufficient = (uint32_t)cp->tlen < OPT_NUM - 1 ? (uint32_t)cp->tlen : OPT_NUM - 1;
    const uint32_t min_match = (cp->mml == 3) ? 3 : 4;
    uint32_t next3 = hc->next_to_update;
    opt_t last_seq;
    memset(&last_seq, 0, sizeof last_seq);
    opt_rescale(o, istart, n, lvl);
    ip += (ip == base + hc->low);
    while (ip < ilimit) {
        uint32_t cur, last_pos = 0;
        {
            const uint32_t litlen = (uint32_t)(ip - anchor);
            const uint32_t ll0 = !litlen;
            const uint32_t nbm = bt_get_all_matches(matches, hc, cp, base, &next3, ip, iend, rep, ll0, min_match);
            if (!nbm) { ip++; continue; }
            opt[0].rep[0] = rep[0]; opt[0].rep[1] = rep[1]; opt[0].rep[2] = rep[2];
            opt[0].mlen = 0;
            opt[0].litlen = litlen;
            opt[0].price = (int)opt_ll_price(o, litlen, lvl);
            {
                const uint32_t max_ml = matches[nbm - 1].len, max_off = matches[nbm - 1].off;
                if (max_ml > sufficient) {
                    last_seq.litlen = litlen; last_seq.mlen = max_ml; last_seq.off = max_off;
                    cur = 0;
                    last_pos = litlen + max_ml;
                    goto shortest_path;
                }
            }
            {
                const uint32_t lit_price = (uint32_t)opt[0].price + opt_ll_price(o, 0, lvl);
                uint32_t pos, k;
                for (pos = 1; pos < min_match; pos++) opt[pos].price = OPT_MAX_PRICE;
                for (k = 0; k < nbm; k++) {
                    const uint32_t off = matches[k].off, end = matches[k].len;
                    for (; pos <= end; pos++) {
                        opt[pos].mlen = pos; opt[pos].off = off; opt[pos].litlen = litlen;
                        opt[pos].price = (int)(lit_price + opt_match_price(o, off, pos, lvl));
                    }
                }
                last_pos = pos - 1;
            }
        }
        for (cur = 1; cur <= last_pos; cur++) {
            const uint8_t *const inr = ip + cur;
            {
                const uint32_t litlen = (opt[cur - 1].mlen == 0) ? opt[cur - 1].litlen + 1 : 1;
                const int price = opt[cur - 1].price + (int)opt_lit_cost1(o, ip[cur - 1], lvl) + (int)opt_ll_price(o, litlen, lvl)
                                  - (int)opt_ll_price(o, litlen - 1, lvl);
                if (price <= opt[cur].price) {
                    opt[cur].mlen = 0; opt[cur].off = 0; opt[cur].litlen = litlen; opt[cur].price = price;
                }
            }
            if (opt[cur].mlen != 0) {
                const uint32_t prev = cur - opt[cur].mlen;
                opt_update_rep(opt[cur].rep, opt[prev].rep, opt[cur].off, opt[cur].litlen == 0);
            } else
                memcpy(opt[cur].rep, opt[cur - 1].rep, sizeof opt[cur].rep);
            if (inr > ilimit) continue; /* the last match starts at least 8 bytes before the end */
            if (cur == last_pos) break;
            if (lvl == 0 && opt[cur + 1].price <= opt[cur].price + (int)(BITCOST / 2)) continue; /* btopt skips unpromising positions */
            {
                const uint32_t ll0 = (opt[cur].mlen != 0);
                const uint32_t litlen = (opt[cur].mlen == 0) ? opt[cur].litlen : 0;
                const uint32_t base_price = (uint32_t)opt[cur].price + opt_ll_price(o, 0, lvl);
                const uint32_t nbm = bt_get_all_matches(matches, hc, cp, base, &next3, inr, iend, opt[cur].rep, ll0, min_match);
                uint32_t k;
                if (!nbm) continue;
                {
                    const uint32_t max_ml = matches[nbm - 1].len;
                    if (max_ml > sufficient || cur + max_ml >= OPT_NUM) {
                        last_seq.mlen = max_ml; last_seq.off = matches[nbm - 1].off; last_seq.litlen = litlen;
                        cur -= (opt[cur].mlen == 0) ? opt[cur].litlen : 0; /* may wrap: then it is the first sequence */
                        last_pos = cur + last_seq.litlen + last_seq.mlen;
                        if (cur > OPT_NUM) cur = 0;
                        goto shortest_path;
                    }
                }
                for (k = 0; k < nbm; k++) {
                    const uint32_t off = matches[k].off, last_ml = matches[k].len;
                    const uint32_t start_ml = k > 0 ? matches[k - 1].len + 1 : min_match;
                    uint32_t mlen;
                    for (mlen = last_ml; mlen >= start_ml; mlen--) {
                        const uint32_t pos = cur + mlen;
                        const int price = (int)(base_price + opt_match_price(o, off, mlen, lvl));
                        if (pos > last_pos || price < opt[pos].price) {
                            while (last_pos < pos) { opt[last_pos + 1].price = OPT_MAX_PRICE; last_pos++; }
                            opt[pos].mlen = mlen; opt[pos].off = off; opt[pos].litlen = litlen; opt[pos].price = price;
                        } else if (lvl == 0)
                            break; /* btopt: early abort of the downward scan */
                    }
                }
            }
        }
        last_seq = opt[last_pos];
        cur = last_pos > last_seq.litlen + last_seq.mlen ? last_pos - (last_seq.litlen + last_seq.mlen) : 0;
    shortest_path:
        if (last_seq.mlen != 0) {
            uint32_t r[3];
            opt_update_rep(r, opt[cur].rep, last_seq.off, last_seq.litlen == 0);
            rep[0] = r[0]; rep[1] = r[1]; rep[2] = r[2];
        } else { rep[0] = opt[cur].rep[0]; rep[1] = opt[cur].rep[1]; rep[2] = opt[cur].rep[2]; }
        {
            const uint32_t store_end = cur + 1;
            uint32_t store_start = store_end, seq_pos = cur, sp;
            opt[store_end] = last_seq;
            while (seq_pos > 0) {
                const uint32_t back = opt[seq_pos].litlen + opt[seq_pos].mlen;
                store_start--;
                opt[store_start] = opt[seq_pos];
                seq_pos = (seq_pos > back) ? seq_pos - back : 0;
            }
            for (sp = store_start; sp <= store_end; sp++) {
                const uint32_t llen = opt[sp].litlen, mlen = opt[sp].mlen, offcode = opt[sp].off;
                if (mlen == 0) { ip = anchor + llen; continue; } /* only literals: the last entry, starts the next stretch */
                opt_update_stats(o, llen, anchor, offcode, mlen);
                store_seq(ss, llen, anchor, offcode, mlen - MINMATCH);
                anchor += llen + mlen;
                ip = anchor;
            }
            opt_set_base(o, lvl);
        }
    }
    return (size_t)(iend - anchor);
}

/* ------------------------------------------------------------ parameters (ZSTD_getCParams) */
static int get_cpar(int level, size_t n, cpar *cp)
{
    /* libzstd 1.4.8's four parameter tables (ZSTD_defaultCParameters: source size > 256 KiB, <= 256 KiB, <= 128 KiB,
     * <= 16 KiB), rows: the base row of the negative levels, then levels 1 .. 22; columns: windowLog, chainLog, hashLog,
     * searchLog, minMatch, targetLength, strategy (1 fast, 2 dfast, 3 greedy, 4 lazy, 5 lazy2, 6 btlazy2, 7 btopt, 8 btultra,
     * 9 btultra2).  Dumped from ZSTD_getCParams and checked against it by the tests. */
    static const int kCParTab[4][23][7] = {
        {{19, 12, 13, 1, 6, 1, 1}, {19, 13, 14, 1, 7, 0, 1}, {20, 15, 16, 1, 6, 0, 1}, {21, 16, 17, 1, 5, 0, 2}, {21, 18, 18, 1, 5, 0, 2}, {21, 18, 19, 2, 5, 2, 3}, {21, 19, 19, 3, 5, 4, 3}, {21, 19, 19, 3, 5, 8, 4}, {21, 19, 19, 3, 5, 16, 5}, {21, 19, 20, 4, 5, 16, 5}, {22, 20, 21, 4, 5, 16, 5}, {22, 21, 22, 4, 5, 16, 5}, {22, 21, 22, 5, 5, 16, 5}, {22, 21, 22, 5, 5, 32, 6}, {22, 22, 23, 5, 5, 32, 6}, {22, 23, 23, 6, 5, 32, 6}, {22, 22, 22, 5, 5, 48, 7}, {23, 23, 22, 5, 4, 64, 7}, {23, 23, 22, 6, 3, 64, 8}, {23, 24, 22, 7, 3, 256, 9}, {25, 25, 23, 7, 3, 256, 9}, {26, 26, 24, 7, 3, 512, 9}, {27, 27, 25, 9, 3, 999, 9}},
        {{18, 12, 13, 1, 5, 1, 1}, {18, 13, 14, 1, 6, 0, 1}, {18, 14, 14, 1, 5, 0, 2}, {18, 16, 16, 1, 4, 0, 2}, {18, 16, 17, 2, 5, 2, 3}, {18, 18, 18, 3, 5, 2, 3}, {18, 18, 19, 3, 5, 4, 4}, {18, 18, 19, 4, 4, 4, 4}, {18, 18, 19, 4, 4, 8, 5}, {18, 18, 19, 5, 4, 8, 5}, {18, 18, 19, 6, 4, 8, 5}, {18, 18, 19, 5, 4, 12, 6}, {18, 19, 19, 7, 4, 12, 6}, {18, 18, 19, 4, 4, 16, 7}, {18, 18, 19, 4, 3, 32, 7}, {18, 18, 19, 6, 3, 128, 7}, {18, 19, 19, 6, 3, 128, 8}, {18, 19, 19, 8, 3, 256, 8}, {18, 19, 19, 6, 3, 128, 9}, {18, 19, 19, 8, 3, 256, 9}, {18, 19, 19, 10, 3, 512, 9}, {18, 19, 19, 12, 3, 512, 9}, {18, 19, 19, 13, 3, 999, 9}},
        {{17, 12, 12, 1, 5, 1, 1}, {17, 12, 13, 1, 6, 0, 1}, {17, 13, 15, 1, 5, 0, 1}, {17, 15, 16, 2, 5, 0, 2}, {17, 17, 17, 2, 4, 0, 2}, {17, 16, 17, 3, 4, 2, 3}, {17, 17, 17, 3, 4, 4, 4}, {17, 17, 17, 3, 4, 8, 5}, {17, 17, 17, 4, 4, 8, 5}, {17, 17, 17, 5, 4, 8, 5}, {17, 17, 17, 6, 4, 8, 5}, {17, 17, 17, 5, 4, 8, 6}, {17, 18, 17, 7, 4, 12, 6}, {17, 18, 17, 3, 4, 12, 7}, {17, 18, 17, 4, 3, 32, 7}, {17, 18, 17, 6, 3, 256, 7}, {17, 18, 17, 6, 3, 128, 8}, {17, 18, 17, 8, 3, 256, 8}, {17, 18, 17, 10, 3, 512, 8}, {17, 18, 17, 5, 3, 256, 9}, {17, 18, 17, 7, 3, 512, 9}, {17, 18, 17, 9, 3, 512, 9}, {17, 18, 17, 11, 3, 999, 9}},
        {{14, 12, 13, 1, 5, 1, 1}, {14, 14, 15, 1, 5, 0, 1}, {14, 14, 15, 1, 4, 0, 1}, {14, 14, 15, 2, 4, 0, 2}, {14, 14, 14, 4, 4, 2, 3}, {14, 14, 14, 3, 4, 4, 4}, {14, 14, 14, 4, 4, 8, 5}, {14, 14, 14, 6, 4, 8, 5}, {14, 14, 14, 8, 4, 8, 5}, {14, 15, 14, 5, 4, 8, 6}, {14, 15, 14, 9, 4, 8, 6}, {14, 15, 14, 3, 4, 12, 7}, {14, 15, 14, 4, 3, 24, 7}, {14, 15, 14, 5, 3, 32, 8}, {14, 15, 15, 6, 3, 64, 8}, {14, 15, 15, 7, 3, 256, 8}, {14, 15, 15, 5, 3, 48, 9}, {14, 15, 15, 6, 3, 128, 9}, {14, 15, 15, 7, 3, 256, 9}, {14, 15, 15, 8, 3, 256, 9}, {14, 15, 15, 8, 3, 512, 9}, {14, 15, 15, 9, 3, 512, 9}, {14, 15, 15, 10, 3, 999, 9}}};
    const int (*t)[7];
    int row, srclog, strat;
    if (level == 0) level = 3;
    if (level > 22 || level < -131072) return -1;
    t = kCParTab[n > 256u * 1024u ? 0 : (n > 128u * 1024u ? 1 : (n > 16u * 1024u ? 2 : 3))];
    row = level < 0 ? 0 : level;
    cp->wlog = t[row][0]; cp->clog = t[row][1]; cp->hlog = t[row][2]; cp->slog = t[row][3]; cp->mml = t[row][4];
    cp->tlen = level < 0 ? -level : t[row][5];
    strat = t[row][6];
    cp->dfast = strat == 2;
    cp->bt = strat >= 6;
    cp->opt = strat >= 7 ? strat - 6 : 0;
    cp->lazy_depth = strat >= 7 ? -1 : (strat == 6 ? 2 : (strat >= 3 ? strat - 3 : -1));
    /* ZSTD_adjustCParams_internal: shrink the window (and hash, chain) to the source size */
    srclog = (n < 64) ? 6 : hb((uint32_t)(n - 1)) + 1;
    if (cp->wlog > srclog) cp->wlog = srclog;
    if (cp->hlog > cp->wlog + 1) cp->hlog = cp->wlog + 1;
    if (cp->clog - cp->bt > cp->wlog) cp->clog = cp->wlog + cp->bt; /* cycleLog = chainLog, minus one for the binary tree (two links per node) */
    if (cp->wlog < 10) cp->wlog = 10;
    if (cp->hlog > 21 || cp->clog > 21) return -1; /* beyond this restatement's static tables (sources above 2 MiB at the deep levels) */
    return 0;
}

/* ------------------------------------------------------------ frame */
size_t cryo_oracle_zstd_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, int level)
{
    static uint32_t table[1 << 21], tshort[1 << 21]; /* hash (long) table; short table / chain table / binary tree */
    static uint32_t table3[1 << 17];                 /* ZSTD_HASHLOG3_MAX */
    static opt_stats ostats;
    hc_state hc;
    static seq_t seqs[ZBLOCK_MAX / 3 + 8];
    static uint8_t lits[ZBLOCK_MAX + 8];
    cpar cp;
    size_t op = 0, ip = 0;
    uint32_t rep[3] = {1, 4, 8};
    uint32_t dict_limit = 1;
    const uint8_t *base = src - 1; /* moves once for btultra2 (below) */
    int first = 1;
    static huf_state hprev, hnext;
    static fse_state fprev, fnext;
    hprev.mode = 0;
    fprev.rep_ll = fprev.rep_of = fprev.rep_ml = 0;
    if (get_cpar(level, n, &cp) || cap < cryo_oracle_zstd_bound(n)) return 0;
    memset(table, 0, sizeof(uint32_t) << cp.hlog);
    if (cp.dfast || cp.lazy_depth >= 0 || cp.opt) memset(tshort, 0, sizeof(uint32_t) << cp.clog);
    g_strategy = cp.bt ? 6 + cp.opt : (cp.lazy_depth >= 0 ? 3 + cp.lazy_depth : (cp.dfast ? 2 : 1));
    hc.hash = table; hc.chain = tshort; hc.next_to_update = 1;
    hc.hash3 = table3; hc.hlog3 = cp.wlog < 17 ? cp.wlog : 17; hc.low = 1;
    if (cp.opt && cp.mml == 3) memset(table3, 0, sizeof(uint32_t) << hc.hlog3);
    ostats.ll_sum = 0; /* ZSTD_reset_matchState: a frame starts without statistics */
    /* frame header: content size always, no checksum, no dictionary id */
    {
        const uint64_t wsize = 1ull << cp.wlog;
        const int single = wsize >= n;
        const int fcs = (n >= 256) + (n >= 65536 + 256) + (n >= 0xFFFFFFFFu);
        dst[0] = 0x28; dst[1] = 0xB5; dst[2] = 0x2F; dst[3] = 0xFD;
        dst[4] = (uint8_t)((single << 5) + (fcs << 6));
        op = 5;
        if (!single) dst[op++] = (uint8_t)((cp.wlog - 10) << 3);
        if (fcs == 0) { if (single) dst[op++] = (uint8_t)n; }
        else if (fcs == 1) { dst[op++] = (uint8_t)(n - 256); dst[op++] = (uint8_t)((n - 256) >> 8); }
        else { dst[op++] = (uint8_t)n; dst[op++] = (uint8_t)(n >> 8); dst[op++] = (uint8_t)(n >> 16); dst[op++] = (uint8_t)(n >> 24); }
    }
    if (n == 0) { dst[op++] = 1; dst[op++] = 0; dst[op++] = 0; return op; }
    while (ip < n) {
        const size_t bs = (n - ip < ZBLOCK_MAX) ? n - ip : ZBLOCK_MAX;
        const int last = (ip + bs == n);
        size_t csize = 0;
        /* libzstd 1.4.8's ZSTD_compress_frameChunk only calls ZSTD_checkDictValidity here (no dictionary: a
         * no-op); window.dictLimit stays 1 for the whole frame and the window is enforced per block through
         * ZSTD_getLowestPrefixIndex.  (Raising dict_limit like ZSTD_window_enforceMaxDist gives the same prefix
         * index but clips repeat offsets too early: caught by tests/stress_gpu.py on a 1 MiB block.) */
        if (bs >= 3 + 3 + 1) {
            seqstore ss;
            uint32_t nrep[3];
            size_t last_ll;
            ss.seqs = seqs; ss.nseq = 0; ss.lits = lits; ss.nlit = 0; ss.long_pos = 0; ss.long_kind = 0;
            nrep[0] = rep[0]; nrep[1] = rep[1]; nrep[2] = rep[2];
            if (cp.lazy_depth >= 0 || cp.opt) {
                /* ZSTD_buildSeqStore: limited catch-up after a very long match */
                const uint32_t cur = (uint32_t)(src + ip - base);
                if (cur > hc.next_to_update + 384u) {
                    const uint32_t d = cur - hc.next_to_update - 384u;
                    hc.next_to_update = cur - (d < 192u ? d : 192u);
                }
            }
            if (cp.opt) {
                if (cp.opt == 3 && ostats.ll_sum == 0 && ip == 0 && bs > 1024) {
                    /* ZSTD_compressBlock_btultra2 -> ZSTD_initStats_ultra: the first block of a frame is parsed twice.  The
                     * first pass only collects statistics; it is then forgotten by moving the window base, so that every
                     * index it left in the tables lies below the lowest valid one */
                    uint32_t trep[3];
                    trep[0] = rep[0]; trep[1] = rep[1]; trep[2] = rep[2];
                    block_opt(&hc, &cp, &ostats, base, src + ip, bs, trep, &ss, 2);
                    ss.nseq = 0; ss.nlit = 0; ss.long_pos = 0; ss.long_kind = 0;
                    base -= bs;
                    hc.low += (uint32_t)bs;
                    hc.next_to_update = hc.low;
                    ostats.lit_sum = opt_upscale(ostats.lit, 255);
                    ostats.ll_sum = opt_upscale(ostats.ll, MaxLL);
                    ostats.ml_sum = opt_upscale(ostats.ml, MaxML);
                    ostats.of_sum = opt_upscale(ostats.of, MaxOff);
                }
                last_ll = block_opt(&hc, &cp, &ostats, base, src + ip, bs, nrep, &ss, cp.opt == 1 ? 0 : 2);
            } else if (cp.lazy_depth >= 0) {
                last_ll = block_lazy(&hc, &cp, base, src + ip, bs, nrep, &ss);
            } else
                last_ll = cp.dfast ? block_dfast(table, tshort, &cp, base, src + ip, bs, nrep, &ss, dict_limit)
                                   : block_fast(table, &cp, base, src + ip, bs, nrep, &ss, dict_limit);
            memcpy(lits + ss.nlit, src + ip + bs - last_ll, last_ll);
            ss.nlit += last_ll;
            csize = compress_sequences(dst + op + 3, seqs, ss.nseq, lits, ss.nlit, bs, ss.long_pos, ss.long_kind, &hprev, &hnext,
                                       &fprev, &fnext, cp.lazy_depth < 0 && !cp.dfast && !cp.opt && cp.tlen > 0);
            if (!first && csize < 25) { /* RLE block for constant non-first blocks */
                size_t k = 1;
                while (k < bs && src[ip + k] == src[ip]) k++;
                if (k == bs) { csize = 1; dst[op + 3] = src[ip]; }
            }
            if (csize > 1) { rep[0] = nrep[0]; rep[1] = nrep[1]; rep[2] = nrep[2]; hprev = hnext; fprev = fnext; }
        }
        if (csize == 0) {
            const uint32_t h = (uint32_t)last + (0u << 1) + ((uint32_t)bs << 3);
            dst[op] = (uint8_t)h; dst[op + 1] = (uint8_t)(h >> 8); dst[op + 2] = (uint8_t)(h >> 16);
            memcpy(dst + op + 3, src + ip, bs);
            op += 3 + bs;
        } else {
            const uint32_t h = csize == 1 ? (uint32_t)last + (1u << 1) + ((uint32_t)bs << 3)
                                          : (uint32_t)last + (2u << 1) + ((uint32_t)csize << 3);
            dst[op] = (uint8_t)h; dst[op + 1] = (uint8_t)(h >> 8); dst[op + 2] = (uint8_t)(h >> 16);
            op += 3 + csize;
        }
        ip += bs;
        first = 0;
    }
    return op;
}
