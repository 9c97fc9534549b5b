/*
 * cryo_synth.h -- specification of the synthetic cryo-block generator.
 *
 * A synthetic block is a valid pg_cryogen block payload as laid out by
 * cryo_init_page()/cryo_storage_insert() (reference storage.c:15-50,
 * storage.h:73-86):
 *
 *   [0,4)  lower  (LE u32)   end of the item-id array
 *   [4,8)  upper  (LE u32)   start of the tuple area
 *   [8,lower)                CryoItemId{u32 off; u32 len} per tuple, 1-based pos
 *   [lower,upper)            zero gap (memset in cryo_init_page, storage.c:18)
 *   [upper,B)                tuples, packed downward from B, each slot
 *                            MAXALIGN(t_len) bytes, pad bytes zero
 *
 * at most CRYO_SYNTH_MAX_TUPLES (=290, storage.c:10,33) tuples per block.
 *
 * Every byte of a block is a pure function of (seed, block_index, block_size,
 * distribution, byte offset), so the HIP generator (one lane per 16 bytes) and
 * the CPU generator produce identical bytes without any sequential state.
 * This header is plain C99 and is included by both the device code and the
 * oracle's CPU generator; it contains the spec, not a codec.
 */
#ifndef CRYO_SYNTH_H
#define CRYO_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define CRYO_HD __host__ __device__ static inline
#else
#define CRYO_HD static inline
#endif

#define CRYO_SYNTH_MAX_TUPLES 290u
#define CRYO_SYNTH_HDR 8u            /* CryoDataHeaderSize, storage.h:86 */
#define CRYO_SYNTH_ITEMID 8u         /* sizeof(CryoItemId), storage.h:73-77 */
#define CRYO_SYNTH_TUPHDR 24u        /* MAXALIGN(SizeofHeapTupleHeader=23) */

/* distributions */
enum {
    CRYO_DIST_WIDE = 0,   /* 290 JSON-like rows filling the block (headline) */
    CRYO_DIST_NARROW = 1, /* 290 x (int4, 32-hex text): t_len 61, ~84% zero gap */
    CRYO_DIST_INT4 = 2,   /* 290 x (int4): t_len 28 (BASELINE config 1 shape)  */
    CRYO_DIST_RANDOM = 3, /* wide-sized tuples, all payload bytes PRNG         */
    CRYO_DIST_ZEROS = 4,  /* empty block: header only                          */
    CRYO_DIST_COUNT = 5
};

/* ---- counter-based PRNG: splitmix64 finaliser over a mixed counter ---- */
CRYO_HD uint64_t cryo_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

CRYO_HD uint64_t cryo_rng(uint64_t seed, uint64_t block, uint32_t tuple, uint32_t ctr)
{
    uint64_t k = 0x9E3779B97F4A7C15ull * (seed + 1u);
    k ^= cryo_mix64(block + 0xD1B54A32D192ED03ull);
    k += ((uint64_t)tuple << 32) | ctr;
    return cryo_mix64(k);
}

/* ---- geometry of a distribution at a block size ---- */
typedef struct {
    uint32_t ntup;     /* tuples in the block */
    uint32_t t_len;    /* HeapTuple t_len     */
    uint32_t slot;     /* MAXALIGN(t_len)     */
    uint32_t lower;    /* end of item ids     */
    uint32_t upper;    /* start of tuple area */
} cryo_synth_geom;

CRYO_HD cryo_synth_geom cryo_synth_geometry(uint32_t B, int dist)
{
    cryo_synth_geom g;
    uint32_t t_len;
    uint32_t ntup = CRYO_SYNTH_MAX_TUPLES;
    switch (dist) {
    case CRYO_DIST_NARROW: t_len = CRYO_SYNTH_TUPHDR + 4u + 1u + 32u; break;
    case CRYO_DIST_INT4:   t_len = CRYO_SYNTH_TUPHDR + 4u; break;
    case CRYO_DIST_ZEROS:  t_len = 0u; ntup = 0u; break;
    default: /* WIDE, RANDOM: rows sized so 290 of them fill the block */
        t_len = ((B - CRYO_SYNTH_HDR - ntup * CRYO_SYNTH_ITEMID) / ntup) & ~7u;
        break;
    }
    g.t_len = t_len;
    g.slot = (t_len + 7u) & ~7u;
    /* small blocks: keep only as many tuples as fit (storage.c:32) */
    while (ntup > 0u && CRYO_SYNTH_HDR + ntup * (CRYO_SYNTH_ITEMID + g.slot) > B) ntup--;
    g.ntup = ntup;
    g.lower = CRYO_SYNTH_HDR + ntup * CRYO_SYNTH_ITEMID;
    g.upper = B - ntup * g.slot;
    return g;
}

/*
 * Text template of a WIDE row: a JSON-like object made of fixed keys and
 * "holes".  Template characters: '#' = hex digit hole, '%' = base64 character
 * hole, '@' = decimal digit
 * hole, '$' = one byte of an 8-byte dictionary word hole (8 consecutive '$'),
 * anything else is literal.  The template is the field pattern below repeated
 * and cut to the text length, so it is computable per byte.
 */
#define CRYO_SYNTH_PERIOD 126u
CRYO_HD uint8_t cryo_synth_tmpl(uint32_t j)
{
    /* 126-byte period: three fields of 42 bytes, each "<key>": "<32 value chars>", */
    const char *p =
        "\"ka\": \"################################\", "   /* 42: 32 hex (md5/uuid-like)     */
        "\"kb\": \"$$$$$$$$-@@@@@@@-###############\", "    /* 42: word, decimal id, 15 hex   */
        "\"kc\": \"################################\", ";  /* 42: 32 hex                      */
    return (uint8_t)p[j % CRYO_SYNTH_PERIOD];
}

CRYO_HD uint8_t cryo_synth_word(uint32_t w, uint32_t k)
{
    const char *words =
        "alpha___bravo___charlie_delta___echo____foxtrot_golf____hotel___"
        "india___juliet__kilo____lima____mike____novembr_oscar___papa____";
    return (uint8_t)words[(w & 15u) * 8u + (k & 7u)];
}

/* byte `j` of the text of tuple `tup` (WIDE) */
CRYO_HD uint8_t cryo_synth_text_byte(uint64_t seed, uint64_t block, uint32_t tup, uint32_t j,
                                     uint32_t textlen)
{
    uint8_t c;
    if (j == 0u) return (uint8_t)'{';
    if (j == textlen - 1u) return (uint8_t)'}';
    c = cryo_synth_tmpl(j - 1u);
    if (c == '#') {
        uint64_t r = cryo_rng(seed, block, tup, j >> 4);
        uint32_t nib = (uint32_t)(r >> ((j & 15u) * 4u)) & 15u;
        return (uint8_t)"0123456789abcdef"[nib];
    }
    if (c == '%') {
        uint64_t r = cryo_rng(seed, block, tup, 0x50000u + (j >> 3));
        uint32_t v = (uint32_t)(r >> ((j & 7u) * 8u)) & 63u;
        return (uint8_t)"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/"[v];
    }
    if (c == '@') {
        uint64_t r = cryo_rng(seed, block, tup, 0x10000u + (j >> 3));
        uint32_t d = (uint32_t)(r >> ((j & 7u) * 8u)) & 255u;
        return (uint8_t)('0' + (d * 10u >> 8));
    }
    if (c == '$') {
        /* position within the 8-byte word hole: the hole spans period offsets 49..56 */
        uint32_t q = (j - 1u) % CRYO_SYNTH_PERIOD;       /* 49..56 */
        uint32_t rep = (j - 1u) / CRYO_SYNTH_PERIOD;
        uint64_t r = cryo_rng(seed, block, tup, 0x20000u + rep);
        return cryo_synth_word((uint32_t)(r >> 17), q - 49u);
    }
    return c;
}

/* byte `k` (0-based, k < t_len) of tuple `tup` (0-based) */
CRYO_HD uint8_t cryo_synth_tuple_byte(uint64_t seed, uint64_t block, uint32_t tup, uint32_t k,
                                      uint32_t t_len, int dist)
{
    uint32_t rowid = (uint32_t)(block * CRYO_SYNTH_MAX_TUPLES + tup + 1u);
    if (dist == CRYO_DIST_RANDOM) {
        uint64_t r = cryo_rng(seed, block, tup, 0x30000u + (k >> 3));
        return (uint8_t)(r >> ((k & 7u) * 8u));
    }
    if (k < CRYO_SYNTH_TUPHDR) {
        /* HeapTupleHeaderData: xmin 0..3, xmax 4..7, cid 8..11, ctid 12..17
         * (bi_hi, bi_lo, posid), infomask2 18..19, infomask 20..21, hoff 22 */
        uint32_t natts = (dist == CRYO_DIST_INT4) ? 1u : 2u;
        uint32_t infomask = (dist == CRYO_DIST_INT4) ? 0x0800u : 0x0802u;
        uint32_t bl = (uint32_t)block + 1u;
        switch (k) {
        case 0: return 0xE8; case 1: return 0x03;     /* xmin = 1000 */
        case 12: return (uint8_t)(bl >> 16); case 13: return (uint8_t)(bl >> 24);
        case 14: return (uint8_t)bl;         case 15: return (uint8_t)(bl >> 8);
        case 16: return (uint8_t)(tup + 1u); case 17: return (uint8_t)((tup + 1u) >> 8);
        case 18: return (uint8_t)natts;
        case 20: return (uint8_t)infomask;   case 21: return (uint8_t)(infomask >> 8);
        case 22: return (uint8_t)CRYO_SYNTH_TUPHDR;
        default: return 0;
        }
    }
    k -= CRYO_SYNTH_TUPHDR;
    if (k < 4u) return (uint8_t)(rowid >> (k * 8u));   /* int4 id */
    k -= 4u;
    if (dist == CRYO_DIST_NARROW) {
        /* 1-byte varlena header ((len+1)<<1|1) then 32 hex chars (md5-like) */
        uint64_t r;
        if (k == 0u) return (uint8_t)(((32u + 1u) << 1) | 1u);
        k -= 1u;
        r = cryo_rng(seed, block, tup, 0x40000u + (k >> 4));
        return (uint8_t)"0123456789abcdef"[(uint32_t)(r >> ((k & 15u) * 4u)) & 15u];
    }
    /* WIDE: 4-byte varlena header (len << 2, LE), then JSON-like text */
    {
        uint32_t textlen = t_len - CRYO_SYNTH_TUPHDR - 4u - 4u;
        uint32_t vl = (textlen + 4u) << 2;
        if (k < 4u) return (uint8_t)(vl >> (k * 8u));
        return cryo_synth_text_byte(seed, block, tup, k - 4u, textlen);
    }
}

/* byte at offset `off` of block `block` */
CRYO_HD uint8_t cryo_synth_byte(uint64_t seed, uint64_t block, uint32_t B, int dist,
                                cryo_synth_geom g, uint32_t off)
{
    if (off < 4u) return (uint8_t)(g.lower >> (off * 8u));
    if (off < 8u) return (uint8_t)(g.upper >> ((off - 4u) * 8u));
    if (off < g.lower) {
        uint32_t i = (off - CRYO_SYNTH_HDR) >> 3, f = (off - CRYO_SYNTH_HDR) & 7u;
        uint32_t v = (f < 4u) ? (B - (i + 1u) * g.slot) : g.t_len;
        return (uint8_t)(v >> ((f & 3u) * 8u));
    }
    if (off < g.upper) return 0;
    {
        /* tuple i occupies [B-(i+1)*slot, B-i*slot) */
        uint32_t fromtop = B - 1u - off;
        uint32_t i = fromtop / g.slot;
        uint32_t k = off - (B - (i + 1u) * g.slot);
        if (k >= g.t_len) return 0;
        return cryo_synth_tuple_byte(seed, block, i, k, g.t_len, dist);
    }
}

#endif /* CRYO_SYNTH_H */
