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
 *   - the first lane with a valid match ends the search; only the probes up to and including it are
 *     committed to the table, colliding ones in ascending order; what the serial walk does behind the match
 *     (store ip - 2, test ip, search on from ip + 1) are further lanes of the same batch when the probes are
 *     consecutive positions: several sequences per batch;
 *   - every probe and candidate is known twelve bytes deep (the four compared and the eight behind them): a match
 *     that ends inside them needs no extension; longer ones go on 64 bytes per step, then 2 KiB per trip;
 *   - what the search does not need is not on its chain: a found sequence (literal run's start, probe, candidate,
 *     match end) waits in a queue, one per lane, and 64 at a time are extended backwards, sized, placed by a scan
 *     and written out lane per sequence (emit_queue).
 * The kernel is latency bound (a block's chain of LDS round trips, and the part of the far candidates' trip to memory
 * that the batch set-up does not cover), so what matters is how short that chain is and how many blocks a CU's LDS
 * holds: the position table is 4096 x (u16 low | 1, 4 or 8 high bits) = 8.5-12 KiB and the ring 1 KiB (measured on
 * 64k x 128 KiB "wide" blocks, round 2: 64 KiB ring 8.6 GB/s, 16 KiB 19, 8 KiB 25.6, 2 KiB 29; round 6:
 * profiles/r06_lz4_enc.txt, 43 -> 68 GB/s).
 */
#include "enc_ring.h"
#include "kernels.h"
#include <cstdio>
#include <cstdlib>

namespace cryo {

namespace {

constexpr bool kLz4EncTagsDefault = false; /* measured: profiles/r06_lz4_enc.txt */
constexpr uint32_t kMfLimit = 12, kLastLiterals = 5, kMinLength = 13, kMaxDist = 65535, kSkipTrigger = 6;

/* position table: u16 low halves + a byte plane per entry = the position's PB high bits (1 up to 128 KiB, 4 up to 1 MiB, 8
 * up to 16 MiB) under a TAG of 8 - PB bits: a hash of the four bytes at that position.
 *
 * Round 6 (profiles/r06_lz4_enc.txt).  A candidate is the LAST position that hashed to the slot: on `wide` 80 % of them lie
 * more than 1 KiB back, outside the ring, and 10.9 such candidates are probed before a sequence's hit
 * (tests/sim_lz4_fast.py).  Their four bytes used to be read from memory in every step -- a trip to memory, with an
 * s_waitcnt vmcnt(0) that also waits for the output stores still on their way --, then the match's extension read the far
 * source again (backward: a second trip, forward: a third).  With the tag a far candidate is read only when its tag equals
 * the probe's (1 in 128 for the accidents; the true match), and the FIRST such lane is taken as the match: its verification
 * word, the 64 bytes before it and the 64 bytes behind its first four ride in ONE trip.  Two thirds of the matches on `wide`
 * are near (median offset 440): those sequences never leave LDS.  (Rounds 1-5 packed the high bits -- 8.5 KiB tables, 15
 * workgroups per CU instead of 11 now; the byte plane also ends the LDS atomics.) */
/* TG: the byte plane with tags (4 KiB); !TG: the position's high bits packed (PB = 1: 512 bytes, 4: 2 KiB; updated with LDS
 * atomics; the owner marks go to the low halves), no tags: every far candidate is read */
template <uint32_t kW, int PB, bool TG>
struct EncLds {
    uint8_t win[kW];
    uint16_t tlo[4096];
    uint8_t thi[TG ? 4096 : 4096 * PB / 8];
};

template <uint32_t kW, uint32_t kStage, int PB, bool TG>
struct Enc : RingIn<kW, kStage> {
    static constexpr uint32_t TB = TG ? 8u - (uint32_t)PB : 0u, PM = (1u << PB) - 1u;
    EncLds<kW, PB, TG> *L;
    uint8_t *dst;
    uint32_t op;
    using RingIn<kW, kStage>::lane;
    using RingIn<kW, kStage>::dw;

    /* LZ4_hash5 of the bytes at p (table log 12): ((v << 24) * 889523592379) >> 52, in 32-bit pieces */
    /* ... and the 8 bytes behind the first four (next8: what a match's forward extension starts with) */
    __device__ inline uint32_t hash(uint32_t p, uint32_t &first4, uint64_t &next8) const
    {
        const uint32_t d0 = dw(p, 0), d1 = dw(p, 1), d2 = dw(p, 2), d3 = dw(p, 3);
        const uint32_t s = p & 3u;
        const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, s);
        const uint32_t b4 = __builtin_amdgcn_ubfe(d1, 8u * s, 8u);
        first4 = lo;
        next8 = (uint64_t)__builtin_amdgcn_alignbyte(d2, d1, s) | ((uint64_t)__builtin_amdgcn_alignbyte(d3, d2, s) << 32);
        const uint32_t x_lo = lo << 24, x_hi = (lo >> 8) | (b4 << 24);
        const uint32_t c_lo = 0x1BBCDCBBu, c_hi = 0xCFu;
        const uint32_t top = __umulhi(x_lo, c_lo) + x_lo * c_hi + x_hi * c_lo;
        return top >> 20;
    }
    static __device__ inline uint32_t tag_of(uint32_t v4) { return TB ? (v4 * 2654435761u) >> (32u - TB) : 0u; }
    /* the slot's high part: (TG) the plane's byte = high bits | tag << PB; (!TG) the packed high bits */
    __device__ inline uint32_t hi_get(uint32_t h) const
    {
        if constexpr (TG) return L->thi[h];
        else if constexpr (PB == 1) return (reinterpret_cast<const uint32_t *>(L->thi)[h >> 5] >> (h & 31u)) & 1u;
        else if constexpr (PB == 4) return (reinterpret_cast<const uint32_t *>(L->thi)[h >> 3] >> (4u * (h & 7u))) & 15u;
        else return L->thi[h];
    }
    __device__ inline void tab_put(uint32_t h, uint32_t v, uint32_t tag)
    {
        L->tlo[h] = (uint16_t)v;
        if constexpr (TG || PB == 8) L->thi[h] = (uint8_t)(((v >> 16) & PM) | (tag << PB));
        else if constexpr (PB == 1) {
            uint32_t *w = reinterpret_cast<uint32_t *>(L->thi) + (h >> 5);
            if ((v >> 16) & 1u) atomicOr(w, 1u << (h & 31u));
            else atomicAnd(w, ~(1u << (h & 31u)));
        } else { /* other lanes of the step may update other nibbles of the word: clear, then set */
            uint32_t *w = reinterpret_cast<uint32_t *>(L->thi) + (h >> 3);
            const uint32_t sh = 4u * (h & 7u);
            atomicAnd(w, ~(15u << sh));
            atomicOr(w, ((v >> 16) & 15u) << sh);
        }
    }

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
        if (from >= this->lo_pos()) { /* the usual run: the ring holds it */
            for (uint32_t i = lane; i < lit; i += 64u) dst[op + i] = (uint8_t)this->byte_ring(from + i);
        } else {
            uint32_t i = lane;
            if (lit >= 2048u) { /* long runs (incompressible blocks are one literal run): 16 bytes per lane, memory to memory */
                const uint32_t whole = lit & ~1023u;
                for (uint32_t o = 16u * lane; o < whole; o += 1024u) {
                    uint4 v;
                    __builtin_memcpy(&v, this->src + from + o, 16);
                    __builtin_memcpy(dst + op + o, &v, 16);
                }
                i += whole;
            }
            for (; i < lit; i += 64u) dst[op + i] = (uint8_t)this->byte_any(from + i);
        }
        op += lit;
    }
};

} // namespace

#ifdef CRYO_LZ4E_PROF /* phase stamps of the encoder (a variant build: profiles/scripts/r06_lz4e3.sh) */
__device__ unsigned long long g_lz4e_prof[16];
#define LZT(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt[k] += t_ - t0; t0 = t_; } while (0)
#else
#define LZT(k) do { } while (0)
#endif

template <uint32_t kW, int PB, bool TG>
__global__ void __launch_bounds__(64)
k_lz4_enc2(const uint8_t *__restrict__ src_base, uint64_t src_stride, uint32_t n, uint64_t n_blocks,
           uint8_t *__restrict__ dst_base, uint64_t dst_stride, int accel_in,
           uint32_t *__restrict__ out_size, int32_t *__restrict__ status, uint32_t dbg)
{
    constexpr uint32_t kStage = kW >= 2048u ? kEncStage : kW / 2u; /* the ring's refill: half of it at most */
    __shared__ __attribute__((aligned(16))) EncLds<kW, PB, TG> L;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t blk = blockIdx.x;
    if (blk >= n_blocks) return;

    Enc<kW, kStage, PB, TG> e;
    using E = Enc<kW, kStage, PB, TG>;
    e.L = &L;
    e.dst = dst_base + uni64(blk * dst_stride);
    e.op = 0;
    const uint32_t accel = accel_in < 1 ? 1u : (accel_in > 65537 ? 65537u : (uint32_t)accel_in);
    const uint8_t *src = src_base + uni64(blk * src_stride);

    for (uint32_t i = lane; i < 512u; i += 64u) reinterpret_cast<uint4 *>(L.tlo)[i] = make_uint4(0, 0, 0, 0);
    if constexpr (TG) {
        /* an empty slot reads as position 0, which is a real candidate (liblz4 puts position 0 into its table before the
         * first probe): every slot starts with the tag of the block's first four bytes */
        uint32_t first4;
        __builtin_memcpy(&first4, src, 4);
        const uint32_t t0 = (E::tag_of(uni(first4)) << PB) & 0xFFu, t4 = t0 * 0x01010101u;
        for (uint32_t i = lane; i < 256u; i += 64u) reinterpret_cast<uint4 *>(L.thi)[i] = make_uint4(t4, t4, t4, t4);
    } else {
        for (uint32_t i = lane; i < sizeof(L.thi) / 16u; i += 64u) reinterpret_cast<uint4 *>(L.thi)[i] = make_uint4(0, 0, 0, 0);
    }
    e.open(L.win, src, n, lane);
    e.ensure(kW);
    __builtin_amdgcn_wave_barrier();
    /* the owner marks (in the byte plane, or -- packed high bits -- in the low halves): every access is one the wave really
     * makes (explicit LDS pointers: a volatile generic one becomes FLAT accesses with a wait behind each) */
    volatile __attribute__((address_space(3))) uint8_t *vthi = (volatile __attribute__((address_space(3))) uint8_t *)L.thi;
    volatile __attribute__((address_space(3))) uint16_t *vtlo = (volatile __attribute__((address_space(3))) uint16_t *)L.tlo;
    constexpr bool kByteMarks = TG || PB == 8;

    const unsigned long long lt_mask = lane ? (~0ull >> (64u - lane)) : 0ull; /* lanes below this one */
    uint32_t anchor = 0;
#ifdef CRYO_LZ4E_PROF
    unsigned long long pt[8] = {0}, t0 = __builtin_amdgcn_s_memtime(), nseq_p = 0, nbatch_p = 0;
#endif

    /* ---- the sequences found wait in a queue, one per lane (literal run's start, the probe that hit, its candidate, the
     * match's end), and are written out 64 at a time.  Nothing the search does next depends on a sequence's bytes in the
     * output, nor on how far its match reaches BACK (the next anchor is the match's end): so the backward extension and the
     * emission leave the block's dependent chain -- they were 1 100 of its 3 850 cycles per sequence on `wide`, wave-uniform
     * work on one sequence -- and are done lane per sequence: 8 bytes before probe and candidate compared at a time; sizes;
     * a scan gives every sequence its place in the output; token, lengths and offset as byte stores, the literals 16 / 8 / 4 /
     * 2 / 1 bytes at a time exact to the byte (memory to memory: the ring has moved on).  A sequence with a long literal run
     * or a very long match is written by the whole wave. ---- */
    uint32_t q_anchor = 0, q_ip = 0, q_match = 0, q_end = 0, qn = 0;
    auto emit_queue = [&]() __attribute__((always_inline)) {
        const bool on = lane < qn;
        /* ONE trip to memory: the 8 bytes before the probe and before its candidate, and the literal run's first 64 bytes
         * (its start does not depend on the backward extension, only its length does) */
        const uint32_t lmax = on ? q_ip - q_anchor : 0u;
        uint64_t bx = 0, by = 1;
        const bool b8 = on && lmax != 0u && q_match >= 8u;
        if (b8) { __builtin_memcpy(&bx, src + q_ip - 8u, 8); __builtin_memcpy(&by, src + q_match - 8u, 8); }
        uint4 lv0 = make_uint4(0, 0, 0, 0), lv1 = lv0, lv2 = lv0, lv3 = lv0;
        {
            const uint8_t *sp = src + q_anchor;
            const bool whole = on && q_anchor + 64u <= n; /* (16-byte loads may reach behind the run, never behind the block) */
            if (whole && lmax > 0u) __builtin_memcpy(&lv0, sp, 16);
            if (whole && lmax > 16u) __builtin_memcpy(&lv1, sp + 16, 16);
            if (whole && lmax > 32u) __builtin_memcpy(&lv2, sp + 32, 16);
            if (whole && lmax > 48u) __builtin_memcpy(&lv3, sp + 48, 16);
            if (on && !whole) { /* the block's last bytes: byte by byte (once per block at most) */
                uint32_t w[16];
#pragma unroll
                for (uint32_t k = 0; k < 16u; k++) w[k] = 0;
                const uint32_t m = lmax < 64u ? lmax : 64u;
#pragma unroll 1
                for (uint32_t k = 0; k < m; k++) {
                    const uint32_t b = (uint32_t)sp[k] << (8u * (k & 3u));
#pragma unroll
                    for (uint32_t j = 0; j < 16u; j++) w[j] |= (k >> 2) == j ? b : 0u;
                }
                lv0 = make_uint4(w[0], w[1], w[2], w[3]); lv1 = make_uint4(w[4], w[5], w[6], w[7]);
                lv2 = make_uint4(w[8], w[9], w[10], w[11]); lv3 = make_uint4(w[12], w[13], w[14], w[15]);
            }
        }
        uint32_t back = 0;
        if (on && lmax != 0u) {
            uint32_t room = lmax < q_match ? lmax : q_match;
            if (b8) {
                const uint64_t d = bx ^ by;
                uint32_t c = d ? (uint32_t)__builtin_clzll(d) >> 3 : 8u;
                c = c < room ? c : room;
                back = c; room = d ? 0u : room - c;
            }
            while (room) { /* rare: more than 8 bytes, or a candidate in the block's first 8 bytes */
                if (q_match - back >= 8u) {
                    uint64_t x, y;
                    __builtin_memcpy(&x, src + q_ip - back - 8u, 8);
                    __builtin_memcpy(&y, src + q_match - back - 8u, 8);
                    const uint64_t d = x ^ y;
                    uint32_t c = d ? (uint32_t)__builtin_clzll(d) >> 3 : 8u;
                    c = c < room ? c : room;
                    back += c; room -= c;
                    if (d) break;
                } else {
                    if (src[q_ip - back - 1u] != src[q_match - back - 1u]) break;
                    back++; room--;
                }
            }
        }
        const uint32_t mstart = q_ip - back;
        const uint32_t lit = on ? mstart - q_anchor : 0u, ml = on ? q_end - (mstart + 4u) : 0u, off = q_ip - q_match;
        const uint32_t nl = lit >= 15u ? (lit - 15u) / 255u + 1u : 0u, nm = ml >= 15u ? (ml - 15u) / 255u + 1u : 0u;
        const uint32_t size = on ? 1u + nl + lit + 2u + nm : 0u;
        const uint32_t incl = scan64_incl(size);
        const uint32_t o0 = e.op + incl - size;
        const bool big = on && (lit > 64u || nm > 4u);
        if (on && !big) {
            uint8_t *d = e.dst + o0;
            *d++ = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (ml < 15u ? ml : 15u));
            if (nl) *d++ = (uint8_t)(lit - 15u); /* lit <= 64: one byte */
            /* the run's bytes from the registers, exact to the byte */
            uint32_t r = lit;
            if (r >= 16u) { __builtin_memcpy(d, &lv0, 16); d += 16; r -= 16u; lv0 = lv1; lv1 = lv2; lv2 = lv3; }
            if (r >= 16u) { __builtin_memcpy(d, &lv0, 16); d += 16; r -= 16u; lv0 = lv1; lv1 = lv2; }
            if (r >= 16u) { __builtin_memcpy(d, &lv0, 16); d += 16; r -= 16u; lv0 = lv1; }
            if (r >= 16u) { __builtin_memcpy(d, &lv0, 16); d += 16; r -= 16u; }
            if (r & 8u) { __builtin_memcpy(d, &lv0.x, 8); d += 8; lv0.x = lv0.z; lv0.y = lv0.w; }
            if (r & 4u) { __builtin_memcpy(d, &lv0.x, 4); d += 4; lv0.x = lv0.y; }
            if (r & 2u) { const uint16_t h2 = (uint16_t)lv0.x; __builtin_memcpy(d, &h2, 2); d += 2; lv0.x >>= 16; }
            if (r & 1u) *d++ = (uint8_t)lv0.x;
            d[0] = (uint8_t)off;
            d[1] = (uint8_t)(off >> 8);
            d += 2;
            if (nm) {
                uint32_t x = ml - 15u;
                for (uint32_t k = 1; k < nm; k++) { *d++ = 255; x -= 255u; }
                *d = (uint8_t)x;
            }
        }
        unsigned long long bm = __ballot(big);
        const uint32_t total = lane_get(incl, 63u);
        while (bm) {
            const uint32_t j = ctz64(bm);
            bm &= bm - 1ull;
            const uint32_t lj = lane_get(lit, j), mj = lane_get(ml, j), oj = lane_get(off, j), aj = lane_get(q_anchor, j);
            e.op = lane_get(o0, j);
            if (lane == 0) e.dst[e.op] = (uint8_t)(((lj < 15u ? lj : 15u) << 4) | (mj < 15u ? mj : 15u));
            e.op++;
            if (lj >= 15u) e.put_len(lj - 15u);
            e.put_literals(aj, lj);
            if (lane == 0) { e.dst[e.op] = (uint8_t)oj; e.dst[e.op + 1] = (uint8_t)(oj >> 8); }
            e.op += 2u;
            if (mj >= 15u) e.put_len(mj - 15u);
        }
        e.op = lane_get(o0, 0u) + total;
    };

    if (n >= kMinLength) {
        const uint32_t mflimit_p1 = n - kMfLimit + 1u;
        const uint32_t matchlimit = n - kLastLiterals;
        /* position 0 goes into the table as index 0: the table is zero already */
        uint32_t ip = 1;
        bool done = false, pre = false; /* pre: a match just ended at ip (table update at ip-2 and re-test at ip pending) */
        uint32_t fwd = ip, step = 1, nb = accel << kSkipTrigger;
        while (!done) {
            /* ================= a batch of 64 probes =================
             * After a match the serial code stores ip-2, then tests ip, then starts the search at ip+1: those two are
             * simply lanes 0 and 1 of the batch (lane order = time order), lane 0 never matching. */
            if (pre) { fwd = ip + 1u; step = 1; nb = accel << kSkipTrigger; }
            const uint32_t sh = pre ? 2u : 0u;
            const uint32_t q = lane - sh;
            /* (the first 64 probes of a search at acceleration 1 are consecutive positions: no scan) */
            const bool fresh = accel == 1u && step == 1u && nb == (1u << kSkipTrigger);
            uint32_t sk, inc;
            if (fresh) { sk = lane < sh ? 0u : 1u; inc = lane < sh ? 0u : q + 1u; }
            else {
                sk = lane < sh ? 0u : (q == 0u ? step : (nb + q - 1u) >> kSkipTrigger);
                inc = scan64_incl(sk);
            }
            uint32_t cur = fwd + inc - sk;
            const uint32_t nxt = fwd + inc;
            if (pre && lane < 2u) cur = lane == 0u ? ip - 2u : ip;
            /* lanes of this batch: until the search would run off the block (prefix-closed: positions only grow), and
             * only as far as the ring can hold next to the first position */
            const bool ends = !(lane < sh || nxt <= mflimit_p1);
            const uint32_t base = pre ? ip - 2u : fwd;
            const bool fits = cur + 12u <= base + (kW - kStage);
            const unsigned long long endm = __ballot(ends);
            const unsigned long long stopm = endm | __ballot(!fits);
            const uint32_t T = stopm ? ctz64(stopm) : 64u; /* lanes 0 .. T-1 take part */
            const bool at_end = T < 64u && ((endm >> T) & 1ull); /* stopped by the block's end, not the ring */
            const bool valid = lane < T;
            if (T == 0u) { done = true; break; }
            /* consecutive positions from lane 1 on (every probe one byte behind the last: the first 64 probes of a search at
             * acceleration 1): what lies behind a match found in this batch can be served from it (below) */
            const bool consecutive = fresh && !(dbg & 4u);
            e.ensure(lane_get(cur, T - 1u) + 12u);
            uint32_t own4 = 0, h = 0, cand = 0, ohi = 0;
            uint64_t own8 = 0;
            bool lost = false;
            if (valid) {
                h = e.hash(cur, own4, own8);
                if constexpr (kByteMarks) {
                    cand = L.tlo[h];
                    ohi = vthi[h];
                    vthi[h] = (uint8_t)lane; /* owner mark; what it overwrites is in ohi and comes back below */
                    lost = vthi[h] != (uint8_t)lane; /* a lane that does not read its own mark back shares the slot */
                    vthi[h] = (uint8_t)ohi;          /* all sharers hold the same old value */
                } else {
                    cand = vtlo[h];
                    ohi = e.hi_get(h);
                    vtlo[h] = (uint16_t)(0xFF00u | lane);
                    lost = vtlo[h] != (uint16_t)(0xFF00u | lane);
                    vtlo[h] = (uint16_t)cand;
                }
            }
            cand |= (ohi & E::PM) << 16;
            /* candidates older than the ring: their four bytes are asked for NOW, before the collisions are sorted out and the
             * near candidates compared -- two LDS round trips the trip to memory hides behind.  (A candidate that the
             * resolution below replaces is a position of this batch, which the ring holds: a far candidate stays what it is.
             * One load per lane and batch serves every search the batch holds.) */
            const uint32_t mytag = E::tag_of(own4);
            uint4 vfar = make_uint4(0, 0, 0, 0); /* 12 of the 16 bytes are used: the four compared, and the eight behind them */
            if (valid && cand + kMaxDist >= cur && cand < e.lo_pos() && (!TG || (ohi >> PB) == mytag || (dbg & 1u)))
                __builtin_memcpy(&vfar, src + cand, 16);
            /* in-batch collisions: an earlier lane with the same hash is what the serial loop would read */
            unsigned long long grouped = 0ull;
            bool resolved = false;
            {
                unsigned long long losers = __ballot(lost);
                while (losers) {
                    const uint32_t j = ctz64(losers);
                    const uint32_t hj = lane_get(h, j);
                    const unsigned long long G = __ballot(valid && h == hj);
                    const unsigned long long below = G & lt_mask;
                    const uint32_t pred = below ? 63u - (uint32_t)__builtin_clzll(below) : lane;
                    const uint32_t pc = (uint32_t)__shfl((int)cur, (int)pred, 64);
                    if (((G >> lane) & 1ull) && below) { cand = pc; resolved = true; }
                    grouped |= G;
                    losers &= ~G;
                }
            }
            /* candidates the ring still holds are compared there; an older one only if its tag says so and it could still
             * be the first hit */
            const bool in_dist = valid && cand + kMaxDist >= cur;
            const bool near = cand >= e.lo_pos();
            /* the candidate's twelve bytes from the ring (whatever a far candidate's lane reads here is not used): four to
             * compare, eight that tell how far a match reaches -- most matches of `wide` end inside them, and need no
             * forward extension of their own */
            uint32_t c4, fq_near;
            {
                const uint32_t c0 = e.dw(cand, 0), c1 = e.dw(cand, 1), c2 = e.dw(cand, 2), c3 = e.dw(cand, 3);
                const uint32_t sc = cand & 3u;
                c4 = __builtin_amdgcn_alignbyte(c1, c0, sc);
                const uint64_t m8 = (uint64_t)__builtin_amdgcn_alignbyte(c2, c1, sc) | ((uint64_t)__builtin_amdgcn_alignbyte(c3, c2, sc) << 32);
                const uint64_t d8 = m8 ^ own8;
                fq_near = d8 ? (uint32_t)__builtin_ctzll(d8) >> 3 : 8u;
            }
            const unsigned long long nearm = __ballot(in_dist && near && c4 == own4);
            const unsigned long long farm = __ballot(in_dist && !near && (!TG || resolved || (ohi >> PB) == mytag || (dbg & 1u)));
#ifdef CRYO_LZ4E_PROF
            nbatch_p++;
#endif
            LZT(0);

            /* ================= the sequences this batch holds =================
             * L0: the lane that only stores its position (ip - 2 behind a match; none in a search that goes on);
             * L1: the first lane that may match.  Lanes L0, L1 .. are the serial walk's next probes. */
            uint32_t L0 = pre ? 0u : 64u, L1 = pre ? 1u : 0u;
            bool next_batch_plain = false;
            for (;;) {
                const unsigned long long from1 = L1 >= 64u ? 0ull : (~0ull << L1);
                unsigned long long hm = nearm & from1;
                bool have_win = false;
                uint32_t wf = 0;
                {
                    const uint32_t first = hm ? ctz64(hm) : 64u;
                    const unsigned long long far = farm & from1 & (first >= 64u ? ~0ull : ((1ull << first) - 1ull));
                    if (far) hm |= __ballot(((far >> lane) & 1ull) && vfar.x == own4);
                }
                const uint32_t K = hm ? ctz64(hm) + 1u : T;
                /* how far the match reaches in the eight bytes behind the four compared (8: further) */
                uint32_t fq = 0;
                if (hm) {
                    if ((farm >> (K - 1u)) & 1ull) {
                        const uint64_t y = (uint64_t)lane_get(vfar.y, K - 1u) | ((uint64_t)lane_get(vfar.z, K - 1u) << 32);
                        const uint64_t x = (uint64_t)lane_get((uint32_t)own8, K - 1u) | ((uint64_t)lane_get((uint32_t)(own8 >> 32), K - 1u) << 32);
                        const uint64_t d8 = x ^ y;
                        fq = d8 ? (uint32_t)__builtin_ctzll(d8) >> 3 : 8u;
                        /* a long match on a far candidate: the next 64 bytes are on their way while the probes are committed */
                        if (fq == 8u && !(dbg & 2u)) {
                            have_win = true;
                            wf = src[lane_get(cand, K - 1u) + 12u + lane]; /* below ip + 12 + lane: a far candidate lies a ring's length back */
                        }
                    } else fq = lane_get(fq_near, K - 1u);
                }
                LZT(1);
                /* commit this search's probes up to K-1: colliding ones one by one, ascending, so the last writer wins */
                {
                    const bool mine = lane < K && (lane >= L1 || lane == L0);
                    if (mine && !((grouped >> lane) & 1ull)) e.tab_put(h, cur, mytag);
                    unsigned long long g = grouped & __ballot(mine);
                    while (g) {
                        const uint32_t j = ctz64(g);
                        if (lane == j) e.tab_put(h, cur, mytag);
                        g &= g - 1ull;
                    }
                }
                LZT(2);
                if (!hm) {
                    if (at_end) { done = true; break; }
                    /* the search goes on behind the batch: the probes counted are the ones behind L1 (or all of a plain batch) */
                    const uint32_t counted = L0 < 64u ? (T > L1 + 1u ? T - L1 - 1u : 0u) : T;
                    fwd = lane_get(nxt, T - 1u);
                    if (counted) { step = (nb + counted - 1u) >> kSkipTrigger; nb += counted; }
                    next_batch_plain = true;
                    break;
                }
                ip = lane_get(cur, K - 1u);
                uint32_t match = lane_get(cand, K - 1u);

                const uint32_t ip_hit = ip;
                LZT(3);
                /* ================= extend forwards: 64 bytes per step, then 2 KiB per step =================
                 * from the four bytes the search compared (what the backward extension added in front of them is equal
                 * already): lane l of the far trip's window is the byte l behind the candidate's first four */
                uint32_t a = ip_hit + 4u + fq, b = match + 12u;
                if (a >= matchlimit) a = matchlimit; /* (the last five bytes of a block are literals) */
                else if (fq == 8u) {
                    bool first = true;
                    for (;;) {
                        e.ensure(a + 64u);
                        const bool inb = a + lane < matchlimit;
                        const uint32_t x = e.byte_ring(a + lane); /* a is a probed position or behind one: the ring holds it */
                        uint32_t y;
                        if (have_win && first) y = wf;
                        else if (b >= e.lo_pos()) y = e.byte_ring(b + lane); /* (staging may have pushed the candidate out since) */
                        else y = inb ? e.byte_any(b + lane) : 1u;
                        const bool eq = inb && x == y;
                        const unsigned long long neq = __ballot(!eq);
                        if (neq) { a += ctz64(neq); break; }
                        a += 64u; b += 64u;
                        first = false;
                        if (a + 2048u + 64u <= matchlimit) { /* a long match: 16 bytes per lane straight from memory */
                            a += count_long(src + a, src + b, src + matchlimit, lane, 0u);
                            break;
                        }
                    }
                }
                LZT(4);
                /* ================= the sequence joins the queue: written out 64 at a time (emit_queue) ================= */
                if (lane == qn) { q_anchor = anchor; q_ip = ip_hit; q_match = match; q_end = a; }
                qn++;
                if (qn == 64u) { LZT(5); emit_queue(); qn = 0; LZT(6); }
                ip = a;
                anchor = ip;
#ifdef CRYO_LZ4E_PROF
                nseq_p++;
#endif
                LZT(5);
                if (ip >= mflimit_p1) { done = true; break; }
                pre = true;
                /* The serial walk now stores ip - 2, tests ip and searches on from ip + 1.  If this batch holds those
                 * positions -- consecutive probes, lane = position - (position of lane 1) + 1 -- their hashes, candidates
                 * and comparisons are done already.  A candidate resolved inside the batch assumed that EVERY earlier lane
                 * stores its position; the lanes inside the match and the one at ip - 1 do not: if any of those shares a
                 * slot with another lane, the batch is not used further. */
                if (!consecutive) break;
                {
                    const uint32_t p1 = lane_get(cur, 1u); /* position of lane 1; lane l >= 1 holds p1 + l - 1 */
                    if (ip < p1 + 2u) break;               /* (ip - 2 must be a lane >= 1) */
                    const uint32_t n1 = ip - p1 + 1u, n0 = n1 - 2u;
                    if (n1 >= T) break;
                    const unsigned long long skipped = (((1ull << n0) - 1ull) & ~((1ull << K) - 1ull)) | (1ull << (n1 - 1u));
                    if (grouped & skipped) break;
                    L0 = n0; L1 = n1;
                }
            }
            if (done) break;
            if (next_batch_plain) pre = false;
        }
    }
    if (qn) { emit_queue(); qn = 0; }
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
#ifdef CRYO_LZ4E_PROF
    if (lane == 0) { for (int k = 0; k < 7; k++) atomicAdd(&g_lz4e_prof[k], pt[k]); atomicAdd(&g_lz4e_prof[8], nseq_p); atomicAdd(&g_lz4e_prof[9], nbatch_p); }
#endif
}

hipError_t launch_lz4_compress_batch64(hipStream_t s, const uint8_t *d_src, uint64_t src_stride,
                                       uint32_t block_size, uint64_t n_blocks, uint8_t *d_dst,
                                       uint64_t dst_stride, int accel, uint32_t *d_out_size, int32_t *d_status)
{
    if (n_blocks > 0x7fffffffull) return hipErrorInvalidValue;
    static const uint32_t dbg = cryo_tuning_env("CRYO_LZ4_ENC_DBG") ? (uint32_t)atoi(cryo_tuning_env("CRYO_LZ4_ENC_DBG")) : 0u; /* bisecting aid */
    static const int wkb = cryo_tuning_env("CRYO_LZ4_ENC_WINDOW") ? atoi(cryo_tuning_env("CRYO_LZ4_ENC_WINDOW")) : 2; /* KiB; tuning aid */
    const dim3 grid((uint32_t)n_blocks), wg(64);
    if (wkb >= 64)
        hipLaunchKernelGGL((k_lz4_enc2<65536, 8, true>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride,
                           accel, d_out_size, d_status, dbg);
    else if (wkb >= 32)
        hipLaunchKernelGGL((k_lz4_enc2<32768, 8, true>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride,
                           accel, d_out_size, d_status, dbg);
    else if (wkb >= 16)
        hipLaunchKernelGGL((k_lz4_enc2<16384, 8, true>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride,
                           accel, d_out_size, d_status, dbg);
    else if (wkb >= 8)
        hipLaunchKernelGGL((k_lz4_enc2<8192, 8, true>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride,
                           accel, d_out_size, d_status, dbg);
    else if (wkb >= 4)
        hipLaunchKernelGGL((k_lz4_enc2<4096, 8, true>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride,
                           accel, d_out_size, d_status, dbg);
    else {
        /* 1 KiB ring; tags (12 workgroups per CU at 128 KiB) or packed high bits without tags (16 per CU): tuning aid CRYO_LZ4_ENC_TAGS */
#define LZ4E_LAUNCH(KW, PBV, TGV) hipLaunchKernelGGL((k_lz4_enc2<KW, PBV, TGV>), grid, wg, 0, s, d_src, src_stride, block_size, n_blocks, d_dst, dst_stride, accel, d_out_size, d_status, dbg)
        static const int tags_env = cryo_tuning_env("CRYO_LZ4_ENC_TAGS") ? atoi(cryo_tuning_env("CRYO_LZ4_ENC_TAGS")) : -1;
        const bool tags = tags_env >= 0 ? tags_env != 0 : kLz4EncTagsDefault;
        const bool ring2k = wkb >= 2 && cryo_tuning_env("CRYO_LZ4_ENC_WINDOW") != nullptr;
        if (block_size <= (128u << 10)) {
            if (ring2k) { if (tags) LZ4E_LAUNCH(2048, 1, true); else LZ4E_LAUNCH(2048, 1, false); }
            else { if (tags) LZ4E_LAUNCH(1024, 1, true); else LZ4E_LAUNCH(1024, 1, false); }
        } else if (block_size <= (1u << 20)) {
            if (ring2k) { if (tags) LZ4E_LAUNCH(2048, 4, true); else LZ4E_LAUNCH(2048, 4, false); }
            else { if (tags) LZ4E_LAUNCH(1024, 4, true); else LZ4E_LAUNCH(1024, 4, false); }
        } else LZ4E_LAUNCH(2048, 8, true);
#undef LZ4E_LAUNCH
    }
#ifdef CRYO_LZ4E_PROF
    {
        unsigned long long h[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lz4e_prof), sizeof h);
        fprintf(stderr, "[lz4 enc] %llu sequences, %llu batches (%.2f sequences per batch); cycles per sequence: batch set-up %.0f  decide + far trip %.0f  commit %.0f  queue %.0f  forward %.0f  rest %.0f  write-out %.0f\n",
                h[8], h[9], (double)h[8] / h[9], (double)h[0] / h[8], (double)h[1] / h[8], (double)h[2] / h[8], (double)h[3] / h[8], (double)h[4] / h[8], (double)h[5] / h[8], (double)h[6] / h[8]);
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lz4e_prof), z, sizeof z);
    }
#endif
    return hipGetLastError();
}

} // namespace cryo
