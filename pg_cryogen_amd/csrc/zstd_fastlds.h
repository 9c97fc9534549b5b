/*
 * zstd_fastlds.h -- the match finder of zstd's `fast` strategy (libzstd 1.4.8 ZSTD_compressBlock_fast, no dictionary)
 * for cryo blocks of up to 128 KiB with a hash table of up to 2^13 entries (level 1 and the negative levels at
 * BASELINE's block size), as a kernel of its own: everything the walk touches per position lives in LDS.  Included
 * by zstd_enc.hip inside its namespace.
 *
 * Replaces the match-finding half of ZSTD_compress(dst, bound, src, B, level) (reference compression.c:102-104);
 * restated for the CPU in oracle/zstd_enc_oracle.c (block_fast).  The entropy half runs afterwards as k_zent
 * (zstd_enc.hip) from the sequences and literals this kernel leaves in the block's workspace slot.
 *
 * Why a second formulation (round 6).  k_zstd_enc keeps the table in global memory and pays three dependent trips
 * to memory per sequence with 4 096 waves in flight (20 GB/s at level 1, 79 x its algorithmic traffic).  The walk
 * is a serial recurrence per block, so throughput = blocks in flight / time per sequence, and what holds the time
 * per sequence up is the length of its dependent chain.  Here the chain is LDS round trips (~64-100 cycles each)
 * instead of trips to memory (~1 us under load):
 *   table    2^hlog x 17 bits: 16-bit low halves + a bit plane (16 + 1 KiB at hashLog 13), zeroed per block;
 *   input    the most recent 2 KiB of the block in a ring, staged 512 bytes at a time with the next chunk's
 *            global load in flight; a position the ring no longer (or not yet) holds is read from global memory;
 *   output   sequences {offset code, literal length | match length} and the literal bytes go to the block's
 *            workspace slot in global memory (fire and forget).
 * 19.5 KiB of LDS per block: eight blocks per CU, 2 048 in flight.
 *
 * A step takes the next LPB / 2 iterations of the library's search loop at once, one POSITION per lane (iteration
 * j looks at ip0 = ip + j * step and ip1 = ip0 + 1: lanes 2j and 2j + 1), against the table as it was before the
 * step.  Every lane writes its position into its slot and reads the slot back: a lane that reads another
 * iteration's position shares its slot with it -- exact, no mark array -- and the step is cut in front of the
 * first iteration that could have seen an earlier iteration's write.  The first iteration that finds anything
 * (repeat offset at ip0 + 2, then the candidate of ip0, then that of ip1) ends the step; lanes behind it put back
 * what they overwrote.  Per sequence the dependent LDS trips are: table (read, mark, read back) -> candidates ->
 * match extension both ways -> tail (the two complementary insertions, the immediate-repeat test, the next
 * step's input).
 *
 * LPB = lanes per block: 64 (one block per wave, group-uniform values in scalar registers) or 32 (two blocks per
 * wave, group-uniform values computed per lane; the two halves diverge like threads of a SIMT program).
 */
#pragma once

constexpr uint32_t kFlRing = 2048u, kFlMask = kFlRing - 1u, kFlChunk = 512u;
constexpr uint32_t kFlAhead = 320u;          /* bytes staged beyond the walk's position when a step starts */
constexpr uint32_t kFlMaxBlock = 128u << 10; /* positions fit 17 bits */
constexpr uint32_t kFlMaxHlog = 13;

/* what the finder leaves at the head of a block's workspace slot */
struct FlHdr { uint32_t nseq, nlit, long_pos, long_kind; };
constexpr size_t kFlHdrBytes = 64;
__host__ __device__ inline uint32_t fl_seq_cap(uint32_t n) { return n / 4u + 16u; } /* a sequence covers at least four bytes */
__host__ __device__ inline size_t fl_lit_off(uint32_t n) { return kFlHdrBytes + (((size_t)fl_seq_cap(n) * 8u + 63u) & ~(size_t)63u); }
__host__ __device__ inline size_t fl_slot_bytes(uint32_t n) { return (fl_lit_off(n) + n + 64u + 255u) & ~(size_t)255u; }
/* LDS per block: low halves, bit plane, ring + the mirror of its first 16 bytes */
__host__ __device__ inline uint32_t fl_lds_bytes(int hlog) { return (2u << hlog) + ((1u << hlog) >> 3) + kFlRing + 16u; }

/* Position order inside a step.  Lanes are matched to positions in DESCENDING order (kFlRev): when several lanes of one
 * LDS store write the same address the highest lane's data stays (measured, profiles/r06_zstd_enc.txt), so the position
 * a shared slot is left with is the EARLIEST one and every later sharer reads an earlier position back: the step is cut
 * exactly in front of the second sharer.  With ascending lanes the first sharer reads a later one back, which only
 * proves that it is clean itself, and the step has to end right behind it.  Either way the cut is correct; the order
 * decides how long the steps are. */
#ifndef CRYO_ZFL_REV
#define CRYO_ZFL_REV 1
#endif
constexpr bool kFlRev = CRYO_ZFL_REV != 0;

template <int LPB>
struct FlGroup {
    uint32_t gl;    /* lane inside the group */
    uint32_t gbase; /* the group's first lane */
    /* rank = place in position order; masks are in lane order */
    __device__ inline uint32_t rank() const { return kFlRev ? (uint32_t)LPB - 1u - gl : gl; }
    static __device__ inline uint32_t lane_of(uint32_t r) { return kFlRev ? (uint32_t)LPB - 1u - r : r; }
    static __device__ inline uint32_t first_rank(uint64_t m) /* m != 0 */
    {
        if constexpr (kFlRev) return (uint32_t)LPB - 1u - (63u - (uint32_t)__builtin_clzll(m));
        else return (uint32_t)__builtin_ctzll(m);
    }
    /* the two lanes of iteration jt */
    static __device__ inline uint32_t pair_bits(uint64_t m, uint32_t jt)
    {
        return (uint32_t)(m >> (kFlRev ? (uint32_t)LPB - 2u - 2u * jt : 2u * jt)) & 3u;
    }
    static __device__ inline bool rank_bit(uint64_t m, uint32_t r) { return (m >> lane_of(r)) & 1ull; }
    __device__ inline uint64_t ballot(bool p) const
    {
        const uint64_t b = wave_ballot(p);
        if constexpr (LPB == 64) return b;
        else return (b >> gbase) & 0xFFFFFFFFull;
    }
    /* value of lane `from` (a group-uniform lane number inside the group) */
    __device__ inline uint32_t bcast(uint32_t v, uint32_t from) const
    {
        if constexpr (LPB == 64) return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)uni(from));
        else return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((gbase + from) << 2), (int)v);
    }
    __device__ inline uint32_t u(uint32_t v) const
    {
        if constexpr (LPB == 64) return uni(v);
        else return v;
    }
};

/* LDS is addressed through explicit address-space pointers: a pointer that the compiler cannot prove to be LDS (a select
 * between ring and memory, a volatile access through a struct) becomes a FLAT access, which also counts on vmcnt */
typedef __attribute__((address_space(3))) uint8_t lds8_t;
typedef __attribute__((address_space(3))) uint32_t lds32_t;
typedef volatile __attribute__((address_space(3))) uint16_t lds16v_t;
/* unaligned LDS accesses of 4, 8 and 16 bytes (exact on gfx950) */
struct __attribute__((packed)) FlP32 { uint32_t v; };
struct __attribute__((packed)) FlP64 { uint64_t v; };
struct __attribute__((packed)) FlP128 { uint32_t x, y, z, w; };
__device__ inline uint32_t lds_rd32(const lds8_t *p) { return reinterpret_cast<const __attribute__((address_space(3))) FlP32 *>(p)->v; }
__device__ inline uint64_t lds_rd64(const lds8_t *p) { return reinterpret_cast<const __attribute__((address_space(3))) FlP64 *>(p)->v; }
/* aligned stores of 8 and 16 bytes */
typedef uint32_t fl_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t fl_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void lds_wr64(lds8_t *p, uint32_t x, uint32_t y)
{
    const fl_u32x2 v = {x, y};
    *reinterpret_cast<__attribute__((address_space(3))) fl_u32x2 *>(p) = v;
}
__device__ inline void lds_wr128(lds8_t *p, uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
    const fl_u32x4 v = {x, y, z, w};
    *reinterpret_cast<__attribute__((address_space(3))) fl_u32x4 *>(p) = v;
}

/* the block's input: a ring of the most recent bytes in LDS, global memory behind it */
template <int LPB>
struct FlIn {
    static constexpr uint32_t kPer = kFlChunk / LPB; /* bytes per lane and chunk: 8 or 16 */
    lds8_t *ring;       /* LDS: kFlRing + 16 */
    const uint8_t *src;
    uint32_t n, lo, hi, floor, gl;
    uint32_t p0, p1, p2, p3; /* the next chunk, on its way */

    __device__ inline void fetch()
    {
        const uint32_t o = hi + gl * kPer;
        p0 = p1 = p2 = p3 = 0;
        if (o + kPer <= n) {
            if constexpr (kPer == 16) { uint4 v; __builtin_memcpy(&v, src + o, 16); p0 = v.x; p1 = v.y; p2 = v.z; p3 = v.w; }
            else { uint2 v; __builtin_memcpy(&v, src + o, 8); p0 = v.x; p1 = v.y; }
        } else if (o < n) { /* the block's last, partial piece: never read past its end */
            for (uint32_t k = 0; k < kPer && o + k < n; k++) {
                const uint32_t b = (uint32_t)src[o + k] << (8u * (k & 3u));
                if (k < 4u) p0 |= b; else if (k < 8u) p1 |= b; else if (k < 12u) p2 |= b; else p3 |= b;
            }
        }
    }
    __device__ inline void push()
    {
        const uint32_t ro = (hi + gl * kPer) & kFlMask;
        if constexpr (kPer == 16) {
            lds_wr128(ring + ro, p0, p1, p2, p3);
            if (ro == 0u) lds_wr128(ring + kFlRing, p0, p1, p2, p3);
        } else {
            lds_wr64(ring + ro, p0, p1);
            if (ro < 16u) lds_wr64(ring + kFlRing + ro, p0, p1);
        }
        hi += kFlChunk;
        lo = hi > kFlRing ? hi - kFlRing : 0u;
        if (lo < floor) lo = floor;
        fetch();
    }
    __device__ inline void open(lds8_t *lds, const uint8_t *s, uint32_t len, uint32_t lane_in_group)
    {
        ring = lds; src = s; n = len; lo = hi = floor = 0; gl = lane_in_group;
        fetch();
    }
    /* stage up to position `ip + kFlAhead`; a walk that jumped beyond what is staged starts the ring again */
    __device__ inline void ensure(uint32_t ip)
    {
        if (ip > hi) { hi = ip & ~(kFlChunk - 1u); lo = floor = hi; fetch(); }
        while (hi < n && hi < ip + kFlAhead) push();
    }
    /* The ring is read whatever the position (any masked address is inside it); a lane whose position it does not hold
     * goes to memory afterwards -- normally no lane does and the branch is skipped.  The empty asm makes the loaded
     * value "used" inside the branch, so the s_waitcnt vmcnt(0) it needs stays there: left to the merge point it is
     * executed by every caller, and there it waits for every store of sequences and literals still on its way (stores
     * count on vmcnt on gfx950) -- 105 such waits made the first build 2.5 us per sequence. */
    __device__ inline uint64_t rd64(uint32_t p) const /* p + 8 <= n */
    {
        uint64_t v = lds_rd64(ring + (p & kFlMask));
        if (!(p >= lo && p + 8u <= hi)) { __builtin_memcpy(&v, src + p, 8); asm volatile("" : "+v"(v)); }
        return v;
    }
    __device__ inline uint32_t rd32(uint32_t p) const /* p + 4 <= n */
    {
        uint32_t v = lds_rd32(ring + (p & kFlMask));
        if (!(p >= lo && p + 4u <= hi)) { __builtin_memcpy(&v, src + p, 4); asm volatile("" : "+v"(v)); }
        return v;
    }
    /* four bytes at any position; bytes at or behind the block's end read as zero */
    __device__ inline uint32_t rd32z(uint32_t p) const
    {
        uint32_t v = lds_rd32(ring + (p & kFlMask));
        if (!(p >= lo && p + 4u <= hi)) {
            v = 0;
            if (p + 4u <= n) __builtin_memcpy(&v, src + p, 4);
            else for (uint32_t k = 0; k < 4u && p + k < n; k++) v |= (uint32_t)src[p + k] << (8u * k);
            asm volatile("" : "+v"(v));
        }
        return v;
    }
    __device__ inline uint32_t rd8(uint32_t p) const /* p < n */
    {
        uint32_t v = ring[p & kFlMask];
        if (!(p >= lo && p < hi)) { v = src[p]; asm volatile("" : "+v"(v)); }
        return v;
    }
};

/* bytes equal from a / c on (a > c), limited by the block's end */
template <int LPB>
__device__ inline uint32_t fl_count_fwd(const FlIn<LPB> &in, const FlGroup<LPB> &g, uint32_t a, uint32_t c)
{
    const uint32_t n = in.n;
    uint32_t done = 0;
    for (;;) {
        const uint32_t pa = a + done + 4u * g.gl;
        uint32_t x = in.rd32z(pa) ^ in.rd32z(pa - (a - c));
        if (pa + 4u > n) x |= pa >= n ? 0xFFFFFFFFu : (0xFFFFFFFFu << (8u * (n - pa)));
        const uint64_t m = g.ballot(x != 0u);
        if (m) {
            const uint32_t f = (uint32_t)__builtin_ctzll(m);
            const uint32_t xf = g.bcast(x, f);
            return done + 4u * f + ((uint32_t)__builtin_ctz(xf) >> 3);
        }
        done += 4u * LPB;
        /* long matches (the zero gap of a cryo block): 16 bytes per lane straight from memory while whole rounds fit */
        while (a + done + 16u * LPB <= n) {
            const uint32_t qa = a + done + 16u * g.gl;
            uint4 va, vc;
            __builtin_memcpy(&va, in.src + qa, 16);
            __builtin_memcpy(&vc, in.src + qa - (a - c), 16);
            const uint32_t x0 = va.x ^ vc.x, x1 = va.y ^ vc.y, x2 = va.z ^ vc.z, x3 = va.w ^ vc.w;
            const uint64_t mm = g.ballot((x0 | x1 | x2 | x3) != 0u);
            if (mm) {
                const uint32_t f = (uint32_t)__builtin_ctzll(mm);
                uint32_t fd = x0 ? ((uint32_t)__builtin_ctz(x0) >> 3) : (x1 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3) : (x2 ? 8u + ((uint32_t)__builtin_ctz(x2) >> 3) : 12u + ((uint32_t)__builtin_ctz(x3 | 0x80000000u) >> 3)));
                fd = g.bcast(fd, f);
                return done + 16u * f + fd;
            }
            done += 16u * LPB;
        }
    }
}

/* bytes equal before a / c, at most lim */
template <int LPB>
__device__ inline uint32_t fl_count_back(const FlIn<LPB> &in, const FlGroup<LPB> &g, uint32_t a, uint32_t c, uint32_t lim)
{
    uint32_t done = 0;
    while (done < lim) {
        const uint32_t k = done + g.gl;
        bool ne = true;
        if (k < lim) ne = in.rd8(a - 1u - k) != in.rd8(c - 1u - k);
        const uint64_t m = g.ballot(ne);
        if (m) return done + (uint32_t)__builtin_ctzll(m);
        done += LPB;
    }
    return lim;
}

/* the table: 16-bit low halves and a bit plane for bit 16 (blocks above 64 KiB) */
struct FlTab {
    lds16v_t *lo;
    lds32_t *bits;
    bool wide; /* positions need 17 bits */
    __device__ inline void insert(uint32_t h, uint32_t p) const /* one lane */
    {
        lo[h] = (uint16_t)p;
        if (wide) {
            const uint32_t bit = 1u << (h & 31u);
            __hip_atomic_fetch_and(&bits[h >> 5], ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (p >> 16) __hip_atomic_fetch_or(&bits[h >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
};

/* copies the literal run [from, from + ll) behind the literals collected so far */
template <int LPB>
__device__ inline void fl_copy_literals(const FlIn<LPB> &in, const FlGroup<LPB> &g, uint8_t *lits, uint32_t from, uint32_t ll)
{
    uint32_t k = g.gl;
    if (ll >= 1024u) { /* long runs (incompressible data): 16 bytes per lane, memory to memory */
        const uint32_t whole = ll & ~(16u * LPB - 1u);
        for (uint32_t o = 16u * g.gl; o < whole; o += 16u * LPB) {
            uint4 v;
            __builtin_memcpy(&v, in.src + from + o, 16);
            __builtin_memcpy(lits + o, &v, 16);
        }
        k += whole;
    }
    for (; k < ll; k += LPB) lits[k] = (uint8_t)in.rd8(from + k);
}

/*
 * ZSTD_compressBlock_fast over one block of n <= 128 KiB that is a frame of its own (window >= n: every earlier
 * position is a valid candidate; repeat offsets start as {1, 4}).  Positions are offsets from the block's first
 * byte; a table entry of 0 is "empty" (position 0 is never inserted and never a valid candidate: the library's
 * index 1 = dictLimit).
 */
template <int LPB, bool PROF = false>
__device__ __attribute__((always_inline)) void fl_find_block(lds8_t *lds, const uint8_t *src, uint32_t n, int hlog, int mls, uint32_t step_size,
                              uint8_t *slot, const FlGroup<LPB> &g, unsigned long long *stats)
{
    uint32_t n_steps = 0, n_iters = 0; /* diagnostic counters (stats != nullptr) */
    /* PROF (CRYO_ZFL_STATS builds): cycles per phase of a step */
    unsigned long long pt[PROF ? 8 : 1] = {0}, t0 = PROF ? __builtin_amdgcn_s_memtime() : 0;
#define FLT(k) do { if constexpr (PROF) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pt[k] += t_ - t0; t0 = t_; } } while (0)
    FlTab tab;
    tab.lo = (lds16v_t *)lds;
    tab.bits = (lds32_t *)(lds + (2u << hlog));
    tab.wide = n > 65536u;
    lds8_t *ring = lds + (2u << hlog) + ((1u << hlog) >> 3);
    {
        const uint32_t quads = ((2u << hlog) + ((1u << hlog) >> 3)) / 16u;
        for (uint32_t i = g.gl; i < quads; i += LPB) lds_wr128(lds + 16u * i, 0, 0, 0, 0);
    }
    FlIn<LPB> in;
    in.open(ring, src, n, g.gl);
    uint2 *seqs = reinterpret_cast<uint2 *>(slot + kFlHdrBytes);
    uint8_t *lits = slot + fl_lit_off(n);
    uint32_t nseq = 0, nlit = 0, long_pos = 0, long_kind = 0;

    const uint32_t iend = n, ilimit = n - 8u;
    uint32_t ip = 1, anchor = 0, off1 = 1, off2 = 0; /* ip0 skips the prefix start; offset_2 = 4 exceeds what lies before it */
    const uint32_t j = g.rank() >> 1, b = g.rank() & 1u;
    uint64_t vn = 0;
    uint32_t vn_ip = 0xFFFFFFFFu; /* vn holds the input of a step that starts at vn_ip with the minimal stride */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    while (ip + 1u < ilimit) {
        in.ensure(ip);
        const uint32_t st = ((ip - anchor) >> 7) + step_size;
        const uint32_t i0 = ip + j * st, pos = i0 + b;
        const bool valid = i0 + 1u < ilimit && ((i0 - anchor) >> 7) + step_size == st;
        uint64_t v8 = 0;
        if (vn_ip == ip && st == step_size) v8 = vn;
        else if (valid) v8 = in.rd64(pos);
        const uint32_t h = valid ? hashs_v(v8, hlog, mls) : 0u;
        if constexpr (PROF) asm volatile("" :: "v"(h));
        FLT(0);
        /* table: the slot as it was, then this position into it, then what it holds now */
        uint32_t old = 0, wbits = 0, rb = 0;
        if (valid) {
            old = tab.lo[h];
            if (tab.wide) wbits = tab.bits[h >> 5];
            tab.lo[h] = (uint16_t)pos;
            rb = tab.lo[h];
        }
        /* repeat offset at ip0 + 2: the odd lane of the iteration holds ip0 + 1 .. ip0 + 8 */
        uint64_t rv = 0;
        const bool rc = valid && b == 1u && off1 > 0u && pos >= off1;
        if (rc) rv = in.rd64(pos - off1);
        const uint32_t old_hi = tab.wide ? (wbits >> (h & 31u)) & 1u : 0u;
        const uint32_t cand = old | (old_hi << 16);
        const bool cok = valid && cand != 0u;
        uint32_t cv = 0;
        if (cok) cv = in.rd32(cand);

        if constexpr (PROF) asm volatile("" :: "v"(cv), "v"(rv), "v"(rb));
        FLT(1);
        /* iterations that saw exactly the serial walk's table */
        const uint64_t vm = g.ballot(valid);
        uint32_t nv = (uint32_t)__builtin_popcountll(vm) >> 1;
        {
            const uint32_t d = (rb - i0) & 0xFFFFu; /* 0 or 1: this iteration's own positions */
            const bool loser = valid && d > 1u;
            const uint64_t lm = g.ballot(loser), em = g.ballot(loser && d >= 0x8000u);
            if (lm) {
                const uint32_t jt = FlGroup<LPB>::first_rank(lm) >> 1;
                const uint32_t cut = jt + (FlGroup<LPB>::pair_bits(em, jt) ? 0u : 1u); /* stays when the slot's other writers all come later */
                if (cut < nv) nv = cut;
            }
        }
        const bool inr = valid && j < nv;
        const bool hit = inr && cok && cv == (uint32_t)v8;
        const bool rephit = inr && rc && (uint32_t)(rv >> 8) == (uint32_t)(v8 >> 8);
        const uint64_t hm = g.ballot(hit), rm = g.ballot(rephit);
        const uint64_t bm = g.ballot(rephit && (((uint32_t)rv ^ (uint32_t)v8) & 0xFFu) == 0u);
        const uint64_t am = hm | rm;
        const uint32_t T = am ? FlGroup<LPB>::first_rank(am) >> 1 : nv - 1u;
        /* lanes behind the last committed iteration put back what they overwrote; then the committed positions,
         * ip0's before ip1's (they may share a slot) */
        const bool commit = valid && j <= T;
        if (valid && !commit) tab.lo[h] = (uint16_t)old;
        if (commit && b == 0u) tab.lo[h] = (uint16_t)pos;
        if (commit && b == 1u) tab.lo[h] = (uint16_t)pos;
        if (tab.wide) {
            const uint32_t hp = (uint32_t)__builtin_amdgcn_mov_dpp((int)h, 0xB1, 0xf, 0xf, true); /* the iteration's other slot */
            if (commit && (pos >> 16) != old_hi && !(b == 0u && hp == h))
                __hip_atomic_fetch_xor(&tab.bits[h >> 5], 1u << (h & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        n_steps++;
        n_iters += T + 1u;
        FLT(2);
        if (!am) {
            ip += nv * st;
            continue;
        }

        const uint32_t cur0 = ip + T * st;
        uint32_t ipm, m, known, blim, offcode;
        if (FlGroup<LPB>::rank_bit(rm, 2u * T + 1u)) {
            const uint32_t bb = FlGroup<LPB>::rank_bit(bm, 2u * T + 1u) ? 1u : 0u; /* ip2[-1] == repMatch[-1] */
            ipm = cur0 + 2u - bb;
            m = ipm - off1;
            known = 4u + bb;
            blim = 0;
            offcode = 0;
        } else {
            const uint32_t which = FlGroup<LPB>::rank_bit(hm, 2u * T) ? 0u : 1u;
            m = g.bcast(cand, FlGroup<LPB>::lane_of(2u * T + which));
            ipm = cur0 + which;
            off2 = off1;
            off1 = ipm - m;
            offcode = off1 + 2u;
            known = 4u;
            const uint32_t la = ipm - anchor;
            blim = la < m ? la : m;
        }
        const uint32_t fwd = fl_count_fwd(in, g, ipm + known, m + known);
        const uint32_t back = blim ? fl_count_back(in, g, ipm, m, blim) : 0u;
        const uint32_t mlen = known + fwd + back, start = ipm - back, ll = start - anchor;
        if constexpr (PROF) asm volatile("" :: "v"(mlen));
        FLT(3);
        fl_copy_literals(in, g, lits + nlit, anchor, ll);
        nlit += ll;
        if (ll > 0xFFFFu) { long_kind = 1; long_pos = nseq; }
        if (mlen - 3u > 0xFFFFu) { long_kind = 2; long_pos = nseq; }
        if (g.gl == 0u) seqs[nseq] = make_uint2(offcode + 1u, (ll & 0xFFFFu) | ((mlen - 3u) << 16));
        nseq++;
        ip = start + mlen;
        anchor = ip;
        FLT(4);
        if (ip <= ilimit) {
            in.ensure(ip);
            /* complementary insertions, then immediate repeats (offset_2) as long as they match */
            const uint64_t va = in.rd64(cur0 + 2u), vb = in.rd64(ip - 2u);
            uint32_t r0 = in.rd32(ip), r1 = off2 > 0u ? in.rd32(ip - off2) : ~r0;
            if (g.gl == 0u) {
                tab.insert(hashs_v(va, hlog, mls), cur0 + 2u);
                tab.insert(hashs_v(vb, hlog, mls), ip - 2u);
            }
            while (r0 == r1) {
                const uint32_t rlen = fl_count_fwd(in, g, ip + 4u, ip + 4u - off2) + 4u;
                const uint32_t t = off2; off2 = off1; off1 = t;
                const uint64_t v = in.rd64(ip);
                if (g.gl == 0u) {
                    tab.insert(hashs_v(v, hlog, mls), ip);
                    seqs[nseq] = make_uint2(1u, (rlen - 3u) << 16);
                }
                if (rlen - 3u > 0xFFFFu) { long_kind = 2; long_pos = nseq; }
                nseq++;
                ip += rlen;
                anchor = ip;
                if (ip > ilimit) break;
                in.ensure(ip);
                r0 = in.rd32(ip);
                r1 = in.rd32(ip - off2);
            }
            /* the next step's input */
            if (ip + 1u < ilimit) {
                const uint32_t q = ip + j * step_size + b;
                vn = q + 8u <= n ? in.rd64(q) : 0ull;
                vn_ip = ip;
                if constexpr (PROF) asm volatile("" :: "v"(vn));
            }
        }
        FLT(5);
    }
    {
        const uint32_t last = iend - anchor;
        fl_copy_literals(in, g, lits + nlit, anchor, last);
        nlit += last;
    }
    if constexpr (PROF) { if (stats && g.gl == 0u) for (int k = 0; k < 6; k++) atomicAdd(&stats[4 + k], pt[k]); }
#undef FLT
    if (stats && g.gl == 0u) { atomicAdd(&stats[0], (unsigned long long)n_steps); atomicAdd(&stats[1], (unsigned long long)n_iters); atomicAdd(&stats[2], (unsigned long long)nseq); }
    if (g.gl == 0u) {
        FlHdr hd;
        hd.nseq = nseq; hd.nlit = nlit; hd.long_pos = long_pos; hd.long_kind = long_kind;
        *reinterpret_cast<FlHdr *>(slot) = hd;
    }
}
