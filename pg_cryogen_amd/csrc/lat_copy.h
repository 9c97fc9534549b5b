/*
 * lat_copy.h -- the byte-parallel execution of LZ sequences for FEW blocks per call: output positions by a prefix sum, literal
 * bytes placed and src[b] = where byte b comes from, pointer jumping until every byte points at a literal, gather.  The
 * back half of lz4_lat.hip (which has the reasoning), shared with the zstd pipeline's few-frames path (zstd_pipe.hip,
 * round 5): once a frame's sequences are explicit (k_zmat), executing them is the same problem as an LZ4 block's.
 *
 * Part of what replaces LZ4_decompress_safe / ZSTD_decompress (reference compression.c:84,116) for the reference's own call
 * shapes: one block per call (cache.c:178), 16 cache slots (cache.c:17).
 */
#ifndef CRYO_LAT_COPY_H
#define CRYO_LAT_COPY_H

#include "lz_common.h"
#include "kernels.h"

namespace cryo {
namespace {

struct LatArgs {
    const uint8_t *src_base;
    const uint64_t *src_off;
    const uint32_t *src_size;
    uint8_t *dst_base;
    uint64_t dst_stride;
    uint32_t B, n_blocks;
    int32_t *status;
    /* the index (lz4_index.hip) */
    const uint16_t *tbl;
    const uint2 *seg;
    uint32_t tbl_cap, cap_s, ext, logS; /* S = 1 << logS walkers (descriptors) per block */
    const uint32_t *ixfailed;           /* k_lz4_few_join: 1 = the block has no index (nullptr: k_lz4_index repairs its own) */
    /* per block */
    uint32_t nmax;      /* sequence slots per block */
    uint32_t *segbase;  /* [n_blocks][S]: first sequence of segment s */
    uint32_t *nseq;     /* [n_blocks] */
    uint32_t *ok;       /* [n_blocks] 1: this path decodes the block */
    uint32_t *done;     /* [n_blocks] 1: decoded here (the batch decoder skips it) */
    uint32_t *pos, *opos, *ll, *lpos, *ml, *off, *nxt; /* [n_blocks][nmax] */
    uint32_t *wgsum;    /* [n_blocks][nmax / 256] */
    uint32_t *src;      /* [n_blocks][B rounded up to 16] */
    uint32_t *changed;  /* [rounds + 1] */
    uint32_t bpad;      /* B rounded up to 4096 */
    /* a second literal space (zstd: the frame's pool of decoded Huffman literals): lpos with bit 31 set counts from
     * pool_base + blk * pool_stride instead of the block's input (nullptr: there is none) */
    const uint8_t *pool_base;
    uint64_t pool_stride;
};

/* a wave per block: exclusive scan of the workgroups' bytes (in place) */
__global__ void __launch_bounds__(64) k_lat_scan(LatArgs A)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t blk = blockIdx.x;
    if (uni(A.ok[blk]) == 0u) return;
    const uint32_t nw = (uni(A.nseq[blk]) + 255u) / 256u;
    uint32_t *w = A.wgsum + blk * (A.nmax / 256u);
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nw; b0 += 64u) {
        const uint32_t v = b0 + lane < nw ? w[b0 + lane] : 0u;
        const uint32_t incl = scan64_incl(v);
        if (b0 + lane < nw) w[b0 + lane] = carry + incl - v;
        carry += lane_get(incl, 63);
    }
    if (carry != A.B && lane == 0u) A.ok[blk] = 0u; /* the block does not decode to B bytes */
}

/* a thread per 16 output bytes: literal bytes and where the match bytes come from */
__global__ void __launch_bounds__(256) k_lat_fill(LatArgs A)
{
    const uint32_t blk = blockIdx.y;
    if (A.ok[blk] == 0u) return;
    const uint32_t b0 = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (b0 >= A.B) return;
    const uint32_t n = A.nseq[blk];
    const uint64_t qb = (uint64_t)blk * A.nmax;
    const uint32_t *opos = A.opos + qb;
    /* the last sequence that starts at or before b0 */
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (opos[mid] <= b0) lo = mid; else hi = mid;
    }
    uint32_t i = lo;
    uint32_t so = opos[i], sl = A.ll[qb + i], sm = A.ml[qb + i], sf = A.off[qb + i], sp = A.lpos[qb + i];
    const uint8_t *sb = A.src_base + A.src_off[blk];
    const uint8_t *pb = A.pool_base ? A.pool_base + (uint64_t)blk * A.pool_stride : sb; /* the second literal space (zstd) */
    uint32_t *srcb = A.src + (uint64_t)blk * A.bpad;
    uint8_t *dst = A.dst_base + (uint64_t)blk * A.dst_stride;
    uint32_t v[4] = {0, 0, 0, 0};
    uint32_t sx[16];
#pragma unroll
    for (uint32_t k = 0; k < 16u; k++) {
        const uint32_t b = b0 + k;
        while (b >= so + sl + sm && i + 1u < n) {
            i++;
            so = opos[i]; sl = A.ll[qb + i]; sm = A.ml[qb + i]; sf = A.off[qb + i]; sp = A.lpos[qb + i];
        }
        uint32_t from = b;
        if (b < A.B) {
            if (b < so + sl) v[k >> 2] |= (uint32_t)((sp >> 31) ? pb : sb)[(sp & 0x7fffffffu) + (b - so)] << (8u * (k & 3u));
            else from = b - sf;
        }
        sx[k] = from;
    }
    if (b0 + 16u <= A.B) *reinterpret_cast<uint4 *>(dst + b0) = make_uint4(v[0], v[1], v[2], v[3]);
    else for (uint32_t k = 0; k < 16u && b0 + k < A.B; k++) dst[b0 + k] = (uint8_t)(v[k >> 2] >> (8u * (k & 3u)));
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++)
        *reinterpret_cast<uint4 *>(srcb + b0 + 4u * k) = make_uint4(sx[4 * k], sx[4 * k + 1], sx[4 * k + 2], sx[4 * k + 3]);
}

/* one round of pointer jumping over every byte of every block this path decodes: THREE hops per round (round 5).  A source
 * only ever moves towards the literal byte it ends at, so a value another thread has already shortened is as good as the old
 * one -- the hops need no synchronisation between them, and a chain of depth d is at most ceil(d / 4) deep after a round:
 * half the launches of one hop per round (a launch costs ~4 us whatever it does, and a call is ~30 of them). */
__global__ void __launch_bounds__(256) k_lat_jump(LatArgs A, uint32_t round)
{
    const uint32_t blk = blockIdx.y;
    if (A.ok[blk] == 0u) return;
    if (round != 0u && A.changed[round - 1u] == 0u) return; /* the round before changed nothing: done */
    const uint32_t b0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (b0 >= A.B) return;
    uint32_t *srcb = A.src + (uint64_t)blk * A.bpad;
    uint4 s4 = *reinterpret_cast<const uint4 *>(srcb + b0);
    uint32_t s[4] = {s4.x, s4.y, s4.z, s4.w};
    uint32_t t[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) t[k] = s[k];
#pragma unroll
    for (uint32_t hop = 0; hop < 3u; hop++) {
        uint32_t u[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) u[k] = t[k] != b0 + k ? srcb[t[k]] : t[k]; /* a byte that points at itself is a literal */
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) t[k] = u[k];
    }
    bool ch = false;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) ch = ch || t[k] != s[k];
    if (ch) *reinterpret_cast<uint4 *>(srcb + b0) = make_uint4(t[0], t[1], t[2], t[3]);
    if (wave_any(ch) && (threadIdx.x & 63u) == 0u) A.changed[round] = 1u; /* a plain store: 65 000 atomics on one word took 0.3 ms a round */
}

/* match bytes from the literal bytes they come from */
__global__ void __launch_bounds__(256) k_lat_gather(LatArgs A)
{
    const uint32_t blk = blockIdx.y;
    if (A.ok[blk] == 0u) return;
    const uint32_t b0 = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (b0 < A.B) {
        const uint32_t *srcb = A.src + (uint64_t)blk * A.bpad;
        uint8_t *dst = A.dst_base + (uint64_t)blk * A.dst_stride;
        uint32_t v[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t k4 = 0; k4 < 4u; k4++) {
            const uint4 s4 = *reinterpret_cast<const uint4 *>(srcb + b0 + 4u * k4);
            const uint32_t s[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++)
                if (b0 + 4u * k4 + k < A.B) v[k4] |= (uint32_t)dst[s[k]] << (8u * k);
        }
        if (b0 + 16u <= A.B) *reinterpret_cast<uint4 *>(dst + b0) = make_uint4(v[0], v[1], v[2], v[3]);
        else for (uint32_t k = 0; k < 16u && b0 + k < A.B; k++) dst[b0 + k] = (uint8_t)(v[k >> 2] >> (8u * (k & 3u)));
    }
    if (blockIdx.x == 0u && threadIdx.x == 0u) {
        A.status[blk] = CRYO_ST_OK;
        A.done[blk] = 1u;
    }
}

} // namespace
} // namespace cryo

#endif
