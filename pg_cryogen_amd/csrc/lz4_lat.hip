/*
 * lz4_lat.hip -- LZ4 block decode for FEW blocks per call: every output byte in parallel.
 *
 * Part of what replaces LZ4_decompress_safe(compressed, out, compressed_size, CRYO_BLCKSZ) (reference
 * compression.c:84) -- for the call shapes the reference itself has: one block per call (pg_cryogen.c:726,
 * cache.c:178), 16 cache slots (cache.c:17).  In the batch decoder (lz4_dec2.hip) a block is one wavefront's serial job:
 * 51 000 sequences of a 1 MiB block take 3 ms however idle the other 255 CUs are, and the copy of one block cannot be
 * cut into independent parts (on tuple data 85 % of the offsets are below 3 KiB: every part waits for the end of the one
 * before it).  What can be done in parallel is to resolve, for every output byte, WHERE ITS VALUE COMES FROM:
 *
 *   1  k_lz4_index with 64 walkers per block (lz4_index.hip) finds the sequences;
 *   2  k_lat_parse: a thread per sequence reads its token (literal length, match length, offset, where the next token
 *      starts) -- and checks everything LZ4_decompress_safe checks on the compressed side;
 *   3  k_lat_scan / k_lat_place: output positions by a prefix sum of the sequence lengths; the checks on the output
 *      side (offsets inside the block, the end-of-block rules), the chain of tokens (each one starts where the one
 *      before says);
 *   4  k_lat_fill: a thread per 16 output bytes finds its sequence (binary search), copies literal bytes from the
 *      stream and writes src[b] = b for them, src[b] = b - offset for match bytes;
 *   5  k_lat_jump, up to log2(B) + 1 rounds of pointer jumping: src[b] = src[src[b]] until every byte points at a
 *      literal byte (a round in which nothing changed ends it: later launches return at once);
 *   6  k_lat_gather: out[b] = out[src[b]].
 *
 * Anything that is not a plain valid block -- a check that fails, offset 0 (liblz4 zero-fills), an index the walkers
 * could not complete, a block of almost only literals (left out of the index) -- leaves the block to the batch decoder,
 * which runs afterwards over the blocks not marked done: verdicts and bytes are its.
 */
#include "lz_common.h"
#include "kernels.h"
#include "lat_copy.h"

namespace cryo {

namespace {

constexpr uint32_t kLatMaxS = 1024;                   /* walkers per block of the index pass, at most */
constexpr uint32_t kMFLimit = 12, kLastLiterals = 5; /* LZ4_decompress_safe's end-of-block rules (lz4.c) */


/* segments of a block exactly as k_lz4_index cuts them (lz4_index.hip) */
__device__ inline void lat_segments(uint32_t cs, uint32_t logS, uint32_t &seff, uint32_t &seglen)
{
    const uint32_t kib = cs >> 10;
    const uint32_t lg = kib ? 31u - (uint32_t)__builtin_clz(kib) : 0u;
    const bool cs_small = lz4_index_one_walker(cs);
    const uint32_t ls_ = cs_small ? 0u : (lg < logS ? lg : logS);
    seff = 1u << ls_;
    seglen = (cs + seff - 1u) >> ls_;
}

/* sequences per segment -> first sequence of every segment, sequences of the block; a workgroup of S lanes per block */
__global__ void __launch_bounds__(1024) k_lat_segs(LatArgs A)
{
    __shared__ uint32_t s_w[17];
    __shared__ uint32_t s_later;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t blk = blockIdx.x, S = 1u << A.logS;
    if (threadIdx.x == 0u) s_later = 0u;
    __syncthreads();
    const uint2 d = A.seg[((uint64_t)blk << A.logS) + threadIdx.x];
    const uint32_t c = (d.x & 0xffffu) + d.y;
    const uint32_t incl = scan64_incl(c);
    if (lane == 63u) s_w[wv] = incl;
    if (threadIdx.x != 0u && c != 0u) s_later = 1u;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t k = 0; k < nw; k++) { before += k < wv ? s_w[k] : 0u; total += s_w[k]; }
    A.segbase[blk * S + threadIdx.x] = before + incl - c;
    uint32_t seff, seglen;
    lat_segments(A.src_size[blk], A.logS, seff, seglen);
    /* the walkers' chains did not all meet (the block's first lane walked all of it: positions more than 64 KiB from
     * the segment's start cannot be told from the 16-bit entries), or the block was left out of the index */
    /* ... or its sequences are 64 bytes and more on average: runs -- the zero gap of narrow rows is ONE match of most of the
     * block --, a memset for the batch decoder and a million-deep chain for pointer jumping (64 x 1 MiB `narrow`: 0.57 ms there,
     * 2.7 ms here) */
    const bool plain = total != 0u && total <= A.nmax && (seff == 1u || s_later != 0u) && seglen < 0xF000u &&
                       (A.ixfailed == nullptr || A.ixfailed[blk] == 0u) && (uint64_t)total * 64u >= A.B;
    if (threadIdx.x == 0u) {
        A.nseq[blk] = total;
        A.ok[blk] = plain ? 1u : 0u;
        A.done[blk] = 0u;
    }
}

/* a thread per sequence: the token */
__global__ void __launch_bounds__(256) k_lat_parse(LatArgs A)
{
    __shared__ uint32_t s_base[kLatMaxS + 1u];
    __shared__ uint32_t s_sum[4];
    __shared__ uint32_t s_ok;
    const uint32_t blk = blockIdx.y;
    const uint32_t n = A.nseq[blk];
    if (blockIdx.x * 256u >= n) return;
    const uint32_t S = 1u << A.logS;
    for (uint32_t t = threadIdx.x; t < S; t += 256u) s_base[t] = A.segbase[blk * S + t];
    if (threadIdx.x == 0u) { s_base[S] = n; s_ok = A.ok[blk]; }
    __syncthreads();
    if (s_ok == 0u) return;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool on = i < n;
    uint32_t len = 0;
    if (on) {
        uint32_t s = 0; /* the last segment that starts at or before i */
        for (uint32_t st = S >> 1; st >= 1u; st >>= 1) s += s_base[s + st] <= i ? st : 0u;
        const uint32_t t = i - s_base[s];
        const uint2 d = A.seg[((uint64_t)blk << A.logS) + s];
        const uint32_t ne = d.x & 0xffffu;
        const uint16_t *row = A.tbl + (uint64_t)blk * A.tbl_cap + s * A.cap_s;
        const uint32_t e16 = t < ne ? row[t] : row[A.ext + (d.x >> 16) + (t - ne)];
        const uint32_t cs = A.src_size[blk];
        uint32_t seff, seglen;
        lat_segments(cs, A.logS, seff, seglen);
        const uint32_t g = s * seglen; /* every entry of segment s lies at or behind it, less than 64 KiB away */
        const uint32_t p = g + ((e16 - g) & 0xffffu);
        const uint8_t *sb = A.src_base + A.src_off[blk];
        bool good = p < cs;
        uint32_t ll = 0, lp = 0, ml = 0, of = 0, nx = 0;
        if (good) {
            /* the literal length, read as LZ4_decompress_safe reads it (lz4.c, oracle/lz4_oracle.c) */
            const uint32_t token = sb[p];
            uint32_t ip = p + 1u;
            ll = token >> 4;
            if (ll == 15u) {
                if (ip + 15u >= cs) good = false;
                else {
                    uint32_t b;
                    do {
                        b = sb[ip++];
                        ll += b;
                        if (ip + 15u >= cs) break;
                    } while (b == 255u && ll < 0x01000000u);
                    if (ll >= 0x01000000u) good = false;
                }
            }
            lp = ip;
            if (good) {
                if ((uint64_t)ip + ll + 8u > cs) { /* the block's last sequence: literals to the end of the stream */
                    if (ip + ll != cs) good = false;
                    nx = cs;
                    ml = 0;
                    of = 0;
                } else {
                    ip += ll;
                    of = (uint32_t)sb[ip] | ((uint32_t)sb[ip + 1u] << 8);
                    ip += 2u;
                    ml = token & 15u;
                    if (ml == 15u) {
                        uint32_t b;
                        do {
                            b = sb[ip++];
                            ml += b;
                            if (ip + 4u >= cs) { good = false; break; }
                        } while (b == 255u && ml < 0x01000000u);
                        if (ml >= 0x01000000u) good = false;
                    }
                    ml += 4u;
                    nx = ip;
                    if (of == 0u) good = false; /* liblz4 fills with zeros: the batch decoder's business */
                    /* a block whose bytes are mostly ONE match (the zero gap of narrow rows) is a memset for the batch decoder
                     * and a million-deep chain for pointer jumping: 64 x 1 MiB `narrow` 0.57 ms there, 2.7 ms here */
                    if (ml >= (A.B >> 3)) good = false;
                }
            }
        }
        if (!good) A.ok[blk] = 0u;
        const uint64_t q = (uint64_t)blk * A.nmax + i;
        A.pos[q] = p; A.ll[q] = ll; A.lpos[q] = lp; A.ml[q] = ml; A.off[q] = of; A.nxt[q] = nx;
        len = good ? ll + ml : 0u;
    }
    /* the workgroup's bytes */
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t incl = scan64_incl(len);
    if (lane == 63u) s_sum[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0u) A.wgsum[blk * (A.nmax / 256u) + blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

/* a thread per sequence: output position, the checks that need it, the chain of tokens */
__global__ void __launch_bounds__(256) k_lat_place(LatArgs A)
{
    __shared__ uint32_t s_sum[4];
    __shared__ uint32_t s_ok;
    const uint32_t blk = blockIdx.y;
    const uint32_t n = A.nseq[blk];
    if (blockIdx.x * 256u >= n) return;
    if (threadIdx.x == 0u) s_ok = A.ok[blk];
    __syncthreads();
    if (s_ok == 0u) return;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool on = i < n;
    const uint64_t q = (uint64_t)blk * A.nmax + i;
    uint32_t ll = 0, ml = 0;
    if (on) { ll = A.ll[q]; ml = A.ml[q]; }
    const uint32_t len = ll + ml;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t incl = scan64_incl(len);
    if (lane == 63u) s_sum[wv] = incl;
    __syncthreads();
    uint32_t before = A.wgsum[blk * (A.nmax / 256u) + blockIdx.x];
    for (uint32_t k = 0; k < wv; k++) before += s_sum[k];
    const uint32_t op = before + incl - len;
    if (on) {
        A.opos[q] = op;
        const uint32_t cs = A.src_size[blk];
        bool good = true;
        if (i == 0u) good = A.pos[q] == 0u;
        if (i + 1u < n) {
            /* not the last sequence: LZ4_decompress_safe must not have taken it for the last one, its match must start
             * inside the output and end at least LASTLITERALS before the end of the block, and the next token starts
             * where this one says */
            good = good && ml != 0u && op + ll + kMFLimit <= A.B && A.off[q] <= op + ll && op + len + kLastLiterals <= A.B &&
                   A.nxt[q] == A.pos[q + 1u];
        } else {
            good = good && ml == 0u && A.nxt[q] == cs && op + ll == A.B;
        }
        if (!good) A.ok[blk] = 0u;
    }
}

struct LatLayout {
    Lz4IndexLayout ix;
    uint32_t nmax, bpad, rounds;
    size_t o_segbase, o_nseq, o_ok, o_done, o_pos, o_opos, o_ll, o_lpos, o_ml, o_off, o_nxt, o_wgsum, o_src, o_changed, bytes;
};

inline size_t al256(size_t v) { return (v + 255u) & ~(size_t)255u; }

LatLayout lat_layout(uint64_t n, uint32_t B)
{
    LatLayout y;
    y.ix = lz4_index_layout_few(n, B);
    {   /* sequence slots per block: what the rows hold, and no more than a block of B bytes can have (three bytes a sequence) */
        const uint64_t rows = ((uint64_t)y.ix.cap_main + y.ix.ext) << y.ix.logS, most = (uint64_t)B / 3u + 256u;
        y.nmax = (uint32_t)(((rows < most ? rows : most) + 255u) & ~(uint64_t)255u);
    }
    y.bpad = (B + 4095u) & ~4095u;
    y.rounds = 2;
    while ((1ull << (2u * (y.rounds - 1u))) < B) y.rounds++; /* a chain is at most B hops long and shrinks four times per round (k_lat_jump) */
    size_t o = al256(y.ix.bytes);
    y.o_segbase = o; o = al256(o + ((size_t)n << y.ix.logS) * 4u);
    y.o_nseq = o; o = al256(o + n * 4u);
    y.o_ok = o; o = al256(o + n * 4u);
    y.o_done = o; o = al256(o + n * 4u);
    const size_t per = (size_t)n * y.nmax * 4u;
    y.o_pos = o; o = al256(o + per + 16u);
    y.o_opos = o; o = al256(o + per);
    y.o_ll = o; o = al256(o + per);
    y.o_lpos = o; o = al256(o + per);
    y.o_ml = o; o = al256(o + per);
    y.o_off = o; o = al256(o + per);
    y.o_nxt = o; o = al256(o + per);
    y.o_wgsum = o; o = al256(o + n * (y.nmax / 256u) * 4u);
    y.o_src = o; o = al256(o + (size_t)n * y.bpad * 4u);
    y.o_changed = o; o = al256(o + (y.rounds + 1u) * 4u);
    y.bytes = o;
    return y;
}

} // namespace

bool lz4_latency_eligible(uint64_t n_blocks, uint32_t block_size)
{
    /* tuning aid (debug builds): up to CRYO_LZ4_FEW_MAX blocks / 4 x that many MiB -- where the byte-parallel form stops paying
     * was measured with it (profiles/r06_lz4_decode_batch_shapes.txt) */
    static const uint64_t max_env = cryo_tuning_env("CRYO_LZ4_FEW_MAX") ? (uint64_t)atoll(cryo_tuning_env("CRYO_LZ4_FEW_MAX")) : 0u;
    const uint64_t max_blocks = max_env ? max_env : 64u, max_bytes = max_env ? (max_env << 22) : (64ull << 20);
    return n_blocks >= 1u && n_blocks <= max_blocks && block_size >= (32u << 10) && block_size <= (2u << 20) &&
           n_blocks * (uint64_t)block_size <= max_bytes;
}

size_t lz4_latency_workspace(uint64_t n_blocks, uint32_t block_size) { return lat_layout(n_blocks, block_size).bytes; }

hipError_t launch_lz4_dec_seq_rest(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off, const uint32_t *d_src_size,
                                   uint8_t *d_dst, uint64_t dst_stride, uint32_t block_size, uint64_t n_blocks, int32_t *d_status,
                                   const void *ws, const Lz4IndexLayout &Lx, const uint32_t *d_done);

hipError_t launch_lz4_decompress_latency(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off,
                                         const uint32_t *d_src_size, uint8_t *d_dst, uint64_t dst_stride,
                                         uint32_t block_size, uint64_t n_blocks, int32_t *d_status, void *d_workspace,
                                         size_t workspace_bytes)
{
    if (n_blocks == 0) return hipSuccess;
    if (!lz4_latency_eligible(n_blocks, block_size) || !d_workspace) return hipErrorInvalidValue;
    const LatLayout y = lat_layout(n_blocks, block_size);
    if (workspace_bytes < y.bytes) return hipErrorInvalidValue;
    uint8_t *ws = static_cast<uint8_t *>(d_workspace);
    if (hipError_t e = launch_lz4_index_few(s, d_src, d_src_off, d_src_size, n_blocks, block_size, d_workspace, y.ix); e != hipSuccess) return e;
    LatArgs A = {};
    A.src_base = d_src; A.src_off = d_src_off; A.src_size = d_src_size;
    A.dst_base = d_dst; A.dst_stride = dst_stride; A.B = block_size; A.n_blocks = (uint32_t)n_blocks; A.status = d_status;
    A.tbl = reinterpret_cast<const uint16_t *>(ws);
    A.seg = reinterpret_cast<const uint2 *>(ws + y.ix.seg_off);
    A.tbl_cap = y.ix.cap; A.cap_s = y.ix.cap_main + y.ix.ext; A.ext = y.ix.ext; A.logS = y.ix.logS;
    A.ixfailed = lz4_index_few_failed(d_workspace, y.ix, n_blocks);
    A.nmax = y.nmax;
    A.segbase = reinterpret_cast<uint32_t *>(ws + y.o_segbase);
    A.nseq = reinterpret_cast<uint32_t *>(ws + y.o_nseq);
    A.ok = reinterpret_cast<uint32_t *>(ws + y.o_ok);
    A.done = reinterpret_cast<uint32_t *>(ws + y.o_done);
    A.pos = reinterpret_cast<uint32_t *>(ws + y.o_pos);
    A.opos = reinterpret_cast<uint32_t *>(ws + y.o_opos);
    A.ll = reinterpret_cast<uint32_t *>(ws + y.o_ll);
    A.lpos = reinterpret_cast<uint32_t *>(ws + y.o_lpos);
    A.ml = reinterpret_cast<uint32_t *>(ws + y.o_ml);
    A.off = reinterpret_cast<uint32_t *>(ws + y.o_off);
    A.nxt = reinterpret_cast<uint32_t *>(ws + y.o_nxt);
    A.wgsum = reinterpret_cast<uint32_t *>(ws + y.o_wgsum);
    A.src = reinterpret_cast<uint32_t *>(ws + y.o_src);
    A.changed = reinterpret_cast<uint32_t *>(ws + y.o_changed);
    A.bpad = y.bpad;
    if (hipError_t e = hipMemsetAsync(A.changed, 0, (y.rounds + 1u) * 4u, s); e != hipSuccess) return e;
    const uint32_t nb = (uint32_t)n_blocks;
    hipLaunchKernelGGL(k_lat_segs, dim3(nb), dim3(1u << y.ix.logS), 0, s, A);
    hipLaunchKernelGGL(k_lat_parse, dim3(y.nmax / 256u, nb), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_lat_scan, dim3(nb), dim3(64), 0, s, A);
    hipLaunchKernelGGL(k_lat_place, dim3(y.nmax / 256u, nb), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_lat_fill, dim3((block_size + 4095u) / 4096u, nb), dim3(256), 0, s, A);
    for (uint32_t r = 0; r < y.rounds; r++)
        hipLaunchKernelGGL(k_lat_jump, dim3((block_size + 1023u) / 1024u, nb), dim3(256), 0, s, A, r);
    hipLaunchKernelGGL(k_lat_gather, dim3((block_size + 4095u) / 4096u, nb), dim3(256), 0, s, A);
    /* what was left (a failed check, offset 0, an index the walkers could not complete, blocks of almost only literals): the
     * decoder that parses in the wave -- it needs no index, and verdicts and bytes are its */
    return launch_lz4_dec_ring(s, d_src, d_src_off, d_src_size, d_dst, dst_stride, block_size, n_blocks, d_status, false, A.done);
}

} // namespace cryo
