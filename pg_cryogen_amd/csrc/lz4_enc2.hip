/*
 * lz4_enc2.hip -- LZ4 block encode, one wavefront per cryo block, 64 probes per step; output bytes
 * identical to liblz4 1.9.3.
 *
 * Replaces LZ4_compress_fast(data, out, CRYO_BLCKSZ, LZ4_compressBound(CRYO_BLCKSZ),
 * lz4_acceleration_guc) (reference compression.c:70-72) for blocks of 65547 bytes .. 16 MiB (liblz4's
 * byU32 table mode); smaller and larger blocks take the serial kernel in lz4_enc.hip.
 *
 * liblz4's greedy parser is a serial recurrence over one hash table: every probe reads the slot an
 * earlier probe may have written.  To keep its exact parse while using the wave:
 *   - the most recent input lives in an LDS ring, staged 1 KiB at a time with the next chunk's global
 *     load already in flight (probe positions, recent candidates, literal copies and match compares read
 *     it); candidates older than the ring are read from global memory (L2), and only when they could
 *     still be the first hit of the batch;
 *   - the next 64 probe positions of the search (steps grow as in LZ4_compress_generic: searchMatchNb
 *     >> skipTrigger) are hashed and looked up by the 64 lanes at once against the table as it was
 *     before the batch; a lane whose hash was already used by an earlier lane of the same batch takes
 *     that lane's position as its candidate instead -- exactly what the serial loop would have read.
 *     Such collisions are detected by marking the slot's high byte with the lane number and reading it
 *     back, and resolved per colliding hash value (rare);
 *   - the first lane with a valid match ends the batch; only the probes up to and including it are
 *     committed to the table, colliding ones in ascending order;
 *   - backward / forward match extension compare 64 bytes per step; literal runs are copied 64 bytes
 *     per step straight to the output.
 * The kernel is latency bound (serial chain of LDS round trips per sequence), so what matters is waves
 * per CU, i.e. LDS per wave: the position table is 4096 x (u16 low | 1, 4 or 8 high bits) = 8.5-12 KiB and the ring only
 * 2 KiB (measured on 64k x 128 KiB "wide" blocks: 64 KiB ring 8.6 GB/s, 16 KiB 19, 8 KiB 25.6, 2 KiB 29).
 */
#include "enc_ring.h"
#include "kernels.h"
#include <cstdlib>

namespace cryo {

namespace {

constexpr uint32_t kMfLimit = 12, kLastLiterals = 5, kMinLength = 13, kMaxDist = 65535, kSkipTrigger = 6;

/* position table: u16 low halves + HB high bits per entry, packed and updated with LDS atomics when HB < 8:
 * HB = 1 (blocks up to 128 KiB) makes the table 8.5 KiB and 15 workgroups fit a CU instead of 11 (32.5 -> 41.5
 * GB/s on the headline blocks); HB = 4 (up to 1 MiB) 10 KiB, 13 per CU; HB = 8: a byte per entry (up to 16 MiB) */
template <uint32_t kW, int HB>
struct EncLds {
    uint8_t win[kW];
    uint16_t tlo[4096];
    uint8_t thi[4096 * HB / 8];
};

template <uint32_t kW, int HB>
struct Enc : RingIn<kW> {
    static constexpr bool BIT = HB < 8; /* owner marks go to the low halves */
    EncLds<kW, HB> *L;
    uint8_t *dst;
    uint32_t op;
    using RingIn<kW>::lane;
    using RingIn<kW>::dw;

    /* LZ4_hash5 of the bytes at p (table log 12): ((v << 24) * 889523592379) >> 52, in 32-bit pieces */
    __device__ inline uint32_t hash(uint32_t p, uint32_t &first4) const
    {
        const uint32_t d0 = dw(p, 0), d1 = dw(p, 1);
        const uint32_t s = p & 3u;
        const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, s);
        const uint32_t b4 = __builtin_amdgcn_ubfe(d1, 8u * s, 8u);
        first4 = lo;
        const uint32_t x_lo = lo << 24, x_hi = (lo >> 8) | (b4 << 24);
        const uint32_t c_lo = 0x1BBCDCBBu, c_hi = 0xCFu;
        const uint32_t top = __umulhi(x_lo, c_lo) + x_lo * c_hi + x_hi * c_lo;
        return top >> 20;
    }
    __device__ inline uint32_t tab_get(uint32_t h) const
    {
        if constexpr (HB == 1) return (uint32_t)L->tlo[h] | (((reinterpret_cast<const uint32_t *>(L->thi)[h >> 5] >> (h & 31u)) & 1u) << 16);
        else if constexpr (HB == 4) return (uint32_t)L->tlo[h] | (((reinterpret_cast<const uint32_t *>(L->thi)[h >> 3] >> (4u * (h & 7u))) & 15u) << 16);
        else return (uint32_t)L->tlo[h] | ((uint32_t)L->thi[h] << 16);
    }
    __device__ inline void tab_put(uint32_t h, uint32_t v)
    {
        L->tlo[h] = (uint16_t)v;
        if constexpr (HB == 1) {
            uint32_t *w = reinterpret_cast<uint32_t *>(L->thi) + (h >> 5);
            if ((v >> 16) & 1u) atomicOr(w, 1u << (h & 31u));
            else atomicAnd(w, ~(1u << (h & 31u)));
        } else if constexpr (HB == 4) { /* other lanes of the step may update other nibbles of the word: clear, then set */
            uint32_t *w = reinterpret_cast<uint32_t *>(L->thi) + (h >> 3);
            const uint32_t sh = 4u * (h & 7u);
            atomicAnd(w, ~(15u << sh));
            atomicOr(w, ((v >> 16) & 15u) << sh);
        } else L->thi[h] = (uint8_t)(v >> 16);
    }
    /* owner marks for the in-step collision test: in the byte plane, or (BIT) in the low half, whose real value
     * is in `cand` and comes back right after the test */
    __device__ inline void mark(uint32_t h, uint32_t lane_id) { if constexpr (BIT) L->tlo[h] = (uint16_t)(0xFF00u | lane_id); else L->thi[h] = (uint8_t)lane_id; }
    __device__ inline bool marked_by(uint32_t h, uint32_t lane_id) const { if constexpr (BIT) return L->tlo[h] == (uint16_t)(0xFF00u | lane_id); else return L->thi[h] == (uint8_t)lane_id; }
    __device__ inline void unmark(uint32_t h, uint32_t cand) { if constexpr (BIT) L->tlo[h] = (uint16_t)cand; else L->thi[h] = (uint8_t)(cand >> 16); }

    /* 255-run length code */
    __device__ inline void put_len(uint32_t len)
    {
        const uint32_t n255 = len / 255u;
        for (uint32_t i = lane; i < n255; i += 64u) dst[op + i] = 255;
        if (lane == 0) dst[op + n255] = (uint8_t)(len - n255 * 255u);
        op += n255 + 1u;
    }
    __device__ inline void put_literals(uint32_t from, uint32_t lit)
    {
        for (uint32_t i = lane; i < lit; i += 64u) dst[op + i] = (uint8_t)this->byte_any(from + i);
        op += lit;
    }
};

} // namespace

template <uint32_t kW, int HB>
__global__ void __launch_bounds__(64)
k_lz4_enc2(const uint8_t *__restrict__ src_base, uint64_t src_stride, uint32_t n, uint64_t n_blocks,
           uint8_t *__restrict__ dst_base, uint64_t dst_stride, int accel_in,
           uint32_t *__restrict__ out_size, int32_t *__restrict__ status)
{
    __shared__ __attribute__((aligned(16))) EncLds<kW, HB> L;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t blk = blockIdx.x;
    if (blk >= n_blocks) return;

    Enc<kW, HB> e;
    e.L = &L;
    e.dst = dst_base + uni64(blk * dst_stride);
    e.op = 0;
    const uint32_t accel = accel_in < 1 ? 1u : (accel_in > 65537 ? 65537u : (uint32_t)accel_in);

    for (uint32_t i = lane; i < 512u; i += 64u) reinterpret_cast<uint4 *>(L.tlo)[i] = make_uint4(0, 0, 0, 0);
    for (uint32_t i = lane; i < sizeof(L.thi) / 16u; i += 64u) reinterpret_cast<uint4 *>(L.thi)[i] = make_uint4(0, 0, 0, 0);
    e.open(L.win, src_base + uni64(blk * src_stride), n, lane);
    e.ensure(2u * kEncStage);
    __builtin_amdgcn_wave_barrier();

    const unsigned long long lt_mask = lane ? (~0ull >> (64u - lane)) : 0ull; /* lanes below this one */
    uint32_t anchor = 0;

    if (n >= kMinLength) {
        const uint32_t mflimit_p1 = n - kMfLimit + 1u;
        const uint32_t matchlimit = n - kLastLiterals;
        /* position 0 goes into the table as index 0: the table is zero already */
        uint32_t ip = 1;
        bool done = false, pre = false; /* pre: a match just ended at ip (table update at ip-2 and re-test at ip pending) */
        while (!done) {
            /* ================= search: batches of 64 probes =================
             * After a match the serial code stores ip-2, then tests ip, then starts the search at ip+1: those
             * two are simply lanes 0 and 1 of the first batch (lane order = time order), lane 0 never matching. */
            uint32_t fwd = pre ? ip + 1u : ip, step = 1, nb = accel << kSkipTrigger;
            uint32_t match = 0;
            for (;;) {
                const uint32_t sh = pre ? 2u : 0u;
                const uint32_t q = lane - sh;
                const uint32_t sk = lane < sh ? 0u : (q == 0u ? step : (nb + q - 1u) >> kSkipTrigger);
                const uint32_t inc = scan64_incl(sk);
                uint32_t cur = fwd + inc - sk;
                const uint32_t nxt = fwd + inc;
                if (pre && lane < 2u) cur = lane == 0u ? ip - 2u : ip;
                /* lanes of this batch: until the search would run off the block (prefix-closed: positions only
                 * grow), and only as far as the ring can hold next to the first position */
                const bool ends = !(lane < sh || nxt <= mflimit_p1);
                const uint32_t base = pre ? ip - 2u : fwd;
                const bool fits = cur + 9u <= base + (kW - kEncStage);
                const unsigned long long stopm = __ballot(ends || !fits);
                const uint32_t T = stopm ? ctz64(stopm) : 64u; /* lanes 0 .. T-1 take part */
                const bool at_end = T < 64u && ((__ballot(ends) >> T) & 1ull); /* stopped by the block's end, not the ring */
                const bool valid = lane < T;
                if (T == 0u) { done = true; break; }
                e.ensure(lane_get(cur, T - 1u) + 9u);
                uint32_t own4 = 0, h = 0, cand = 0;
                if (valid) {
                    h = e.hash(cur, own4);
                    cand = e.tab_get(h);
                    e.mark(h, lane); /* owner mark; what it overwrites is in cand and comes back below */
                }
                /* in-batch collisions: an earlier lane with the same hash is what the serial loop would read.
                 * Every lane marked its slot; a lane that does not read its own mark back shares the slot. */
                unsigned long long grouped = 0ull;
                {
                    /* the read-back must see what the WAVE wrote, not be forwarded from this lane's own store */
                    asm volatile("" ::: "memory");
                    const bool lost = valid && !e.marked_by(h, lane);
                    asm volatile("" ::: "memory");
                    if (valid) e.unmark(h, cand); /* all sharers hold the same old value */
                    unsigned long long losers = __ballot(lost);
                    while (losers) {
                        const uint32_t j = ctz64(losers);
                        const uint32_t hj = lane_get(h, j);
                        const unsigned long long G = __ballot(valid && h == hj);
                        const unsigned long long below = G & lt_mask;
                        const uint32_t pred = below ? 63u - (uint32_t)__builtin_clzll(below) : lane;
                        const uint32_t pc = (uint32_t)__shfl((int)cur, (int)pred, 64);
                        if (((G >> lane) & 1ull) && below) cand = pc;
                        grouped |= G;
                        losers &= ~G;
                    }
                }
                /* candidates still in the ring first; the older ones only if they could be the first hit */
                const bool testable = valid && !(pre && lane == 0u) && cand + kMaxDist >= cur;
                const bool near = cand >= e.lo_pos();
                unsigned long long hm = __ballot(testable && near && e.rd32(cand) == own4);
                {
                    const uint32_t first = hm ? ctz64(hm) : 64u;
                    const unsigned long long far = __ballot(testable && !near) & (first >= 64u ? ~0ull : ((1ull << first) - 1ull));
                    if (far) {
                        bool hf = false;
                        if ((far >> lane) & 1ull) {
                            uint32_t v;
                            __builtin_memcpy(&v, e.src + cand, 4);
                            hf = v == own4;
                        }
                        hm |= __ballot(hf);
                    }
                }
                const uint32_t K = hm ? ctz64(hm) + 1u : T;
                /* commit probes 0 .. K-1: colliding ones one by one, ascending, so the last writer wins */
                if (lane < K && !((grouped >> lane) & 1ull)) e.tab_put(h, cur);
                {
                    unsigned long long g = grouped & (K >= 64u ? ~0ull : ((1ull << K) - 1ull));
                    while (g) {
                        const uint32_t j = ctz64(g);
                        if (lane == j) e.tab_put(h, cur);
                        g &= g - 1ull;
                    }
                }
                if (hm) {
                    ip = lane_get(cur, K - 1u);
                    match = lane_get(cand, K - 1u);
                    break;
                }
                if (at_end) { done = true; break; }
                fwd = lane_get(nxt, T - 1u);
                if (T > sh) { step = (nb + (T - sh) - 1u) >> kSkipTrigger; nb += T - sh; }
                pre = false;
            }
            if (done) break;

            /* ================= extend backwards ================= */
            {
                uint32_t room = ip - anchor < match ? ip - anchor : match;
                while (room) {
                    const bool in = lane < room;
                    const bool eq = in && e.byte_any(ip - 1u - lane) == e.byte_any(match - 1u - lane);
                    const unsigned long long neq = __ballot(!eq);
                    const uint32_t c = neq ? ctz64(neq) : 64u;
                    ip -= c; match -= c;
                    if (c < 64u) break;
                    room -= 64u;
                }
            }

            const uint32_t lit = ip - anchor;
            /* ================= extend forwards: 64 bytes per step ================= */
            uint32_t a = ip + 4u, b = match + 4u;
            for (;;) {
                e.ensure(a + 64u);
                const bool inb = a + lane < matchlimit;
                const bool eq = inb && e.byte_any(a + lane) == e.byte_any(b + lane); /* after a backward extension even ip may precede the ring */
                const unsigned long long neq = __ballot(!eq);
                if (neq) { a += ctz64(neq); break; }
                a += 64u; b += 64u;
            }
            const uint32_t ml = a - (ip + 4u);
            /* ================= emit: token, literal length, literals, offset, match length ================= */
            if (lane == 0) e.dst[e.op] = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (ml < 15u ? ml : 15u));
            e.op++;
            if (lit >= 15u) e.put_len(lit - 15u);
            e.put_literals(anchor, lit);
            if (lane == 0) {
                e.dst[e.op] = (uint8_t)(ip - match);
                e.dst[e.op + 1] = (uint8_t)((ip - match) >> 8);
            }
            e.op += 2u;
            if (ml >= 15u) e.put_len(ml - 15u);
            ip = a;
            anchor = ip;
            if (ip >= mflimit_p1) break;
            pre = true;
        }
    }
    /* ================= last literals ================= */
    {
        const uint32_t lit = n - anchor;
        if (lane == 0) e.dst[e.op] = (uint8_t)((lit < 15u ? lit : 15u) << 4);
        e.op++;
        if (lit >= 15u) e.put_len(lit - 15u);
        e.ensure(n);
        e.put_literals(anchor, lit);
    }
    if (lane == 0) { out_size[blk] = e.op; status[blk] = CRYO_ST_OK; }
}

hipError_t launch_lz4_compress_batch64(hipStream_t s, const uint8_t *d_src, uint64_t src_stride,
                                       uint32_t block_size, uint64_t n_blocks, uint8_t *d_dst,
                                       uint64_t dst_stride, int accel, uint32_t *d_out_size, int32_t *d_status)
{
    if (n_blocks > 0x7fffffffull) return hipErrorInvalidValue;
    static const int wkb = cryo_tuning_env("CRYO_LZ4_ENC_WINDOW") ? atoi(cryo_tuning_env("CRYO_LZ4_ENC_WINDOW")) : 2; /* KiB; tuning aid */
    const dim3 grid((uint32_t)n_blocks), wg(64);
    if (wkb >= 64)
        hipLaunchKernelGGL((k_lz4_enc2<65536, 8>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride,
                           accel, d_out_size, d_status);
    else if (wkb >= 32)
        hipLaunchKernelGGL((k_lz4_enc2<32768, 8>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride,
                           accel, d_out_size, d_status);
    else if (wkb >= 16)
        hipLaunchKernelGGL((k_lz4_enc2<16384, 8>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride,
                           accel, d_out_size, d_status);
    else if (wkb >= 8)
        hipLaunchKernelGGL((k_lz4_enc2<8192, 8>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride,
                           accel, d_out_size, d_status);
    else if (wkb >= 4)
        hipLaunchKernelGGL((k_lz4_enc2<4096, 8>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride,
                           accel, d_out_size, d_status);
    else if (block_size <= (128u << 10))
        hipLaunchKernelGGL((k_lz4_enc2<2048, 1>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride,
                           accel, d_out_size, d_status);
    else if (block_size <= (1u << 20))
        hipLaunchKernelGGL((k_lz4_enc2<2048, 4>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride,
                           accel, d_out_size, d_status);
    else
        hipLaunchKernelGGL((k_lz4_enc2<2048, 8>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride,
                           accel, d_out_size, d_status);
    return hipGetLastError();
}

} // namespace cryo
