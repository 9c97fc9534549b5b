/*
 * lz4_index.hip -- the sequence index of the LZ4 block decoder: where every sequence (token) of a block starts.
 *
 * Part of what replaces LZ4_decompress_safe(compressed, out, compressed_size, CRYO_BLCKSZ) (reference
 * compression.c:84).  Finding the sequence starts of an LZ4 block is a serial walk (token -> literal length -> next
 * token); inside the decoding wave it costs speculative per-byte tables (lz4_dec.hip), across a batch it is
 * embarrassingly parallel.  k_lz4_index runs it with one LANE per walker and writes, per block, a row of 16-bit
 * entries (the low 16 bits of every token's offset in the compressed block) that k_lz4_dec_seq (lz4_dec2.hip) turns
 * into one sequence per lane.  The decoder checks every entry against the stream: a wrong row costs speed, never bytes.
 *
 * The walk.  A lane that read its stream straight from global memory paid a trip to L2 per hop (6.9 ms for the
 * headline batch).  So every walker stages its stream through a private 512-byte LDS ring, filled 128 bytes (one
 * cache line) at a time: in turn j of four the wave's 64 lanes load one 16-byte piece each for 16 walkers that have
 * room, and store it one round later, so a load's latency is covered by four hops.  The walk is a small state
 * machine per lane (token / literal-length extension / match-length extension), one LDS read per hop serves every
 * lane whatever it is in; a token with at most two literals and a short match leaves the next token inside the bytes
 * just read and that one is taken in the same turn.  The pass is paced by (hops per walker) x (time of a turn): all
 * walkers of a batch are in flight at once (one wave per SIMD, LDS-bound), so a block's walk is as long as the pass.
 *
 * Several walkers per block (round 3).  With one walker per block the pass takes as long for 4 096 blocks as for
 * 65 536, and eight times longer for 1 MiB blocks than for 128 KiB ones.  With S walkers per block, walker s starts at
 * a GUESSED token position (byte s * csize / S of the stream) and walks segment s.  Its chain is wrong at first, but two
 * chains that ever visit the same position are identical from there on, and on real data a chain that starts anywhere
 * runs into the true one within a few tokens (a false chain hops ~7 bytes at a time over literal bytes, a true token
 * lies every ~16 bytes).  So after the walk each walker s keeps going past the end of its segment until it meets a
 * position walker s+1 recorded ("extension", a handful of hops, direct reads from memory); what it visited on the way
 * replaces the false start of walker s+1's records.  By induction from walker 0 (whose start is true) every segment is
 * then described by two pieces: the extension of its left neighbour + its own records from the meeting point on.  A
 * boundary where the chains do not meet within the segment (periodic data can do that) makes the block's first lane
 * walk the whole block again, alone: never worse than one walker per block was.
 */
#include "lz_common.h"
#include <cstdlib>

namespace cryo {

constexpr uint32_t kIdxDist = 2; /* rounds between the request of a chunk and its store into the ring (one round of distance: no faster, profiles/r03_variants_ab.txt) */
/* Geometry.  A chunk is what one walker is fed at a time: kIdxLpw lanes x 16 bytes = a cache line.  A turn issues two loads,
 * each serving 64 / kIdxLpw walkers; a round is four turns; 512-byte rings, every walker served once per round -- 39 KB of
 * LDS per wave, one wave per SIMD.  (Round 4 tried two waves per SIMD with 64-byte chunks and 256-byte rings: 2.24 -> 3.09 ms
 * -- a walker uses 0 .. 70 bytes of its stream per turn and runs dry on 256 bytes; the ring a walker needs is set by the
 * variance of its appetite, not by the latency to cover.  profiles/r04_lz4_decode_ab.txt; the variant is
 * profiles/scripts/r06_removed_variants.patch.) */
constexpr uint32_t kIdxLanes = 64, kIdxRing = 512;
constexpr uint32_t kIdxLpw = 8u;                             /* lanes that load one walker's chunk */
constexpr uint32_t kIdxChunk = kIdxLpw * 16u;                /* 128 bytes */
constexpr uint32_t kIdxWpl = kIdxLanes / kIdxLpw;            /* walkers per load: 8 */
constexpr uint32_t kIdxGroups = kIdxLanes / (2u * kIdxWpl);  /* walker groups, one per turn in rotation: 4 */
constexpr uint32_t kIdxLine = 16u;                           /* positions per stored line (32 bytes) */
constexpr uint32_t kIdxPutStores = kIdxLine / 8u;            /* 16-byte stores of IDX_PUT per round */
constexpr uint32_t kIdxTrashPerLane = 1u;                    /* a trash slot per lane */
constexpr uint32_t kIdxStride = kIdxRing + 16u; /* bank skew between rings */

__device__ inline uint32_t bperm(uint32_t v, uint32_t src_lane)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}
/* lane i gets lane i+1's value (lane 63: 0) / lane i-1's (lane 0: 0) */
__device__ inline uint32_t from_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); }
__device__ inline uint32_t from_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }

/* The token after the one at virtual position p, read straight from memory (the extension walk: a handful of hops per
 * walker; every hop of k_lz4_index_few).  x = the four bytes at p (up to 3 bytes beyond the stream: inside the slack every
 * source buffer has), loaded by the caller so that it can put its own stores BEHIND the load: vmcnt counts in order, and a
 * store issued in front of the load is waited for with it -- a trip to memory more on every hop.  Returns a position >= vend
 * when the token at p is the stream's last sequence (or the stream is cut).  255-runs are skipped eight bytes per load (the
 * zero gap of a cryo block is one match of ~100 KB ... 1 MB: 400 ... 4 000 extension bytes). */
__device__ inline uint32_t lz4_skip_255(const uint8_t *sb, const uint32_t vend, uint32_t q, uint32_t *sum)
{
    /* q: first extension byte not looked at yet, known to follow a 255; returns the position behind the terminating byte
     * (0xffffffff: the stream ends first), adds the bytes' values to *sum */
    for (;;) {
        if (q >= vend) return 0xffffffffu;
        if (q + 64u <= vend) { /* a long run: 64 bytes per trip while they are all 255 */
            uint4 a, b, c, d;
            __builtin_memcpy(&a, sb + q, 16); __builtin_memcpy(&b, sb + q + 16u, 16);
            __builtin_memcpy(&c, sb + q + 32u, 16); __builtin_memcpy(&d, sb + q + 48u, 16);
            if ((a.x & a.y & a.z & a.w & b.x & b.y & b.z & b.w & c.x & c.y & c.z & c.w & d.x & d.y & d.z & d.w) == 0xffffffffu) {
                if (sum) { *sum += 255u * 64u; if (*sum >= 0x40000000u) return 0xffffffffu; }
                q += 64u;
                continue;
            }
        }
        if (q + 8u <= vend) {
            unsigned long long v;
            __builtin_memcpy(&v, sb + q, 8);
            const unsigned long long nv = ~v;
            const uint32_t n = nv ? (uint32_t)__builtin_ctzll(nv) >> 3 : 8u; /* leading 0xFF bytes */
            if (sum) *sum += 255u * n;
            if (n < 8u) {
                if (sum) *sum += (uint32_t)(v >> (8u * n)) & 255u;
                return q + n + 1u;
            }
            q += 8u;
            if (sum && *sum >= 0x40000000u) return 0xffffffffu;
        } else {
            const uint32_t b = sb[q++];
            if (sum) *sum += b;
            if (b != 255u) return q;
        }
    }
}
__device__ inline uint32_t lz4_next_token_word(const uint8_t *sb, const uint32_t vend, const uint32_t p, const uint32_t x)
{
    const uint32_t t = x & 255u;
    uint32_t ll = t >> 4, q = p + 1u;
    if (ll == 15u) {
        const uint32_t b = (x >> 8) & 255u;
        ll += b;
        q++;
        if (b == 255u) {
            q = lz4_skip_255(sb, vend, q, &ll);
            if (q == 0xffffffffu || ll >= vend) return 0xffffffffu;
        }
    }
    q += ll;
    if (q + 2u > vend) return 0xffffffffu; /* literals only: the last sequence */
    q += 2u;
    if ((t & 15u) == 15u) {
        if (q >= vend) return 0xffffffffu;
        const uint32_t b = sb[q++];
        if (b == 255u) q = lz4_skip_255(sb, vend, q, nullptr);
    }
    return q;
}
__device__ inline uint32_t lz4_next_token_direct(const uint8_t *sb, const uint32_t vend, const uint32_t p)
{
    uint32_t x;
    __builtin_memcpy(&x, sb + p, 4);
    return lz4_next_token_word(sb, vend, p, x);
}

__global__ void __launch_bounds__(64)
k_lz4_index(const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off,
            const uint32_t *__restrict__ src_size, const uint64_t n_blocks, uint16_t *__restrict__ tbl,
            const uint32_t logS, const uint32_t cap_main, const uint32_t ext, const uint32_t cap,
            uint2 *__restrict__ seg, uint16_t *__restrict__ dummy_base, const uint32_t block_size)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[kIdxLanes * kIdxStride + (kIdxTrashPerLane ? kIdxLanes * 16u : 16u)]; /* + trash */
    /* the last positions of every lane (two lines: one being filled, one waiting for its store) + a slot where a lane
     * that records nothing writes (per lane in the round-3 geometry, one for the wave now) */
    __shared__ __attribute__((aligned(16))) uint16_t s_pos[kIdxLanes][2u * kIdxLine + (kIdxTrashPerLane ? 8u : 0u)];
    __shared__ __attribute__((aligned(16))) uint16_t s_ptrash[8];
    const uint32_t lane = threadIdx.x;
    const uint32_t S = 1u << logS, cap_s = cap_main + ext;
    const uint64_t gl = (uint64_t)blockIdx.x * kIdxLanes + lane;
    const uint64_t blk = gl >> logS;
    const uint32_t sw = (uint32_t)gl & (S - 1u);   /* walker number inside the block */
    const bool owner = blk < n_blocks;
    /* stream of this lane's block, in "virtual" positions: vp = delta + offset in the block, so that chunk
     * addresses are 128-byte aligned */
    uint64_t aoff = 0;
    uint32_t delta = 0, vend = 0, seff = 1, seglen = 0;
    if (owner) {
        const uint64_t o = src_off[blk];
        const uint32_t cs = src_size[blk];
        aoff = o & ~(uint64_t)127;     /* chunks are whole 128-byte lines: each line of the input is fetched once */
        delta = (uint32_t)(o & 127u);
        vend = delta + cs;
        /* segments of at least 1 KiB: a block that compressed to little is walked by fewer lanes */
        const uint32_t kib = cs >> 10;
        const uint32_t lg = kib ? 31u - (uint32_t)__builtin_clz(kib) : 0u;
        const bool cs_small = lz4_index_one_walker(cs);
        const uint32_t ls_ = cs_small ? 0u : (lg < logS ? lg : logS);
        seff = 1u << ls_;
        seglen = (cs + seff - 1u) >> ls_;
    }
    if (!owner) aoff = src_off[0] & ~(uint64_t)127; /* a lane past the end of the batch re-reads block 0 */
    /* with several walkers per block, blocks of almost only literals are not indexed: the in-wave parser decodes them (kernels.h) */
    const bool walker = owner && sw < seff && !(logS != 0u && lz4_literal_heavy(vend - delta, block_size));
    const uint32_t gstart = delta + sw * seglen;                                    /* guessed (walker 0: true) start */
    uint32_t stop = (walker && sw + 1u < seff) ? delta + (sw + 1u) * seglen : vend; /* first position of the next segment */
    uint16_t *const rowbase = tbl + blk * cap;
    uint16_t *row = rowbase + sw * cap_s + ext;           /* this walker's own records */
    uint16_t *const dummy = dummy_base + lane * 16u;      /* 32 bytes per lane behind the rows: where lanes with nothing to store store */
    uint32_t kcap = (logS != 0u && seff == 1u) ? S * cap_s - ext : cap_main; /* a block's only walker has the whole row (as the redo of phase 3 has) */
    uint32_t pos = gstart;                /* next byte to interpret */
    uint32_t requested = gstart & ~(kIdxChunk - 1u); /* chunks requested up to here (multiple of kIdxChunk) */
    uint32_t filled = requested;          /* chunks stored in the ring up to here */
    /* 0 token, 1 literal-length extension, 2 match-length extension.  A walker with a guessed start begins in state 2: if the
     * guess lies inside a run of 255s (the match-length bytes of a cryo block's zero gap: 4 100 of them at 1 MiB, several
     * segments long) it skips to the byte behind the run's end -- a true token there --; read as a token, a 255 would send
     * it through a literal length of megabytes and out of the stream.  Anywhere else state 2 just moves the guess by a byte. */
    uint32_t state = (walker && sw != 0u) ? 2u : 0u;
    uint32_t acc = 0, tm = 0;             /* literal length being accumulated; match nibble of the current token */
    uint32_t k = 0, ls = 0;               /* positions recorded; 16-entry lines of them stored */
    uint16_t *pbuf = s_pos[lane];
    uint16_t *const ptrash = kIdxTrashPerLane ? pbuf + 2u * kIdxLine : s_ptrash;
    bool done = !walker || vend == delta;

    /* what this lane serves: one 16-byte piece of the next chunk of walker (lane / kIdxLpw) of the load's set of kIdxWpl
     * walkers; turn j feeds group j % kIdxGroups = sets 2 (j % kIdxGroups) and 2 (j % kIdxGroups) + 1 */
    const uint32_t piece16 = (lane & (kIdxLpw - 1u)) * 16u;
    const uint32_t wil = lane / kIdxLpw; /* walker inside a load's set */
#define IDX_SRC(q) const uint64_t saoff##q = ((uint64_t)bperm((uint32_t)(aoff >> 32), kIdxWpl * q + wil) << 32) | bperm((uint32_t)aoff, kIdxWpl * q + wil); \
                   const uint32_t svend##q = bperm(vend, kIdxWpl * q + wil);
    IDX_SRC(0) IDX_SRC(1) IDX_SRC(2) IDX_SRC(3)
    IDX_SRC(4) IDX_SRC(5) IDX_SRC(6) IDX_SRC(7)
#undef IDX_SRC
    const uint32_t rb = lane * kIdxStride; /* this lane's ring inside s_ring */

    /* Chunks on their way: two per turn, committed kIdxDist rounds later.  The loads are inline assembly with the
     * waits written by hand (round 4).  Written as plain C++ loads the compiler kept the eight chunks of a round in one
     * set of registers and copied them into a second set at the top of the loop -- and a copy needs the data, so it put
     * s_waitcnt vmcnt(7) ... vmcnt(0) in front of the copies: every round began by draining the loads the previous turn
     * had issued a few hundred cycles earlier, a whole trip to memory exposed per round on a wave that has its SIMD to
     * itself (39 % of the wave's cycles were SQ_WAIT_ANY, profiles/r03_lz4_dec_sq.json).  vmcnt counts in issue order:
     * a round issues 2 stores (IDX_PUT) + 4 x 2 loads, so when turn j commits what it requested kIdxDist rounds ago,
     * exactly 10 * kIdxDist - 2 younger operations may still be in flight.
     * Separate variables, not arrays: the compiler kept an indexed array in scratch memory. */
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr uint32_t kTrash = kIdxLanes * kIdxStride; /* where a slot that asked for nothing commits */
    const uint32_t mytrash = kTrash + (kIdxTrashPerLane ? lane * 16u : 0u);
#define IDX_SLOT(n) u32x4 fd##n = {0, 0, 0, 0}, fe##n = fd##n; uint32_t fa##n = mytrash, fb##n = fa##n; /* nothing requested yet */
    IDX_SLOT(0) IDX_SLOT(1) IDX_SLOT(2) IDX_SLOT(3)
    IDX_SLOT(4) IDX_SLOT(5) IDX_SLOT(6) IDX_SLOT(7)
#undef IDX_SLOT

    /* a chunk's size while a chunk of this lane is on its way, else 0: per slot set (A / B) and per visit of the round
     * (the round-4 geometry serves a walker twice per round: turns g and g + kIdxGroups) */
    uint32_t outA0 = 0, outA1 = 0, outB0 = 0, outB1 = 0;
    auto turn = [&](const uint32_t j, u32x4 &fd, u32x4 &fe, uint32_t &fa, uint32_t &fb, uint32_t &out128,
                    const uint64_t soff, const uint32_t sve, const uint64_t soff2, const uint32_t sve2) __attribute__((always_inline)) {
        const uint32_t grp = j % kIdxGroups;
        const bool myturn = lane / (2u * kIdxWpl) == grp; /* this lane's walker is in the group the turn feeds */
        /* ---- commit what the slot's loads of a round ago brought (to the trash slot if they were idle re-reads) ----
         * Everything per-lane in this turn is evaluated EAGERLY (& and | on the predicates, selects instead of ifs): with
         * && / || / if the compiler built exec-mask branches around one- and two-instruction bodies, 450 scalar mask
         * instructions per round of 1160; a lone wave per SIMD issues one instruction per four cycles whatever its kind,
         * so the pass is as long as its instruction count (round 3: 3.06 -> 2.4 ms for the headline batch). */
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((8 + kIdxPutStores) * (kIdxDist - 1) + 6 + kIdxPutStores) : "memory");
        *reinterpret_cast<u32x4 *>(s_ring + fa) = fd;
        *reinterpret_cast<u32x4 *>(s_ring + fb) = fe;
        {
            const uint32_t got = myturn ? out128 : 0u; /* this lane's chunk, if it asked for one, is in its ring now */
            filled += got;
            out128 -= got;
        }
        /* the hop's ring reads go out before the exchange below: one LDS round trip per turn, not two */
        const uint32_t w0 = *reinterpret_cast<const uint32_t *>(s_ring + rb + (pos & (kIdxRing - 4u)));
        const uint32_t w1 = *reinterpret_cast<const uint32_t *>(s_ring + rb + ((pos + 4u) & (kIdxRing - 4u)));
        const uint32_t w2 = *reinterpret_cast<const uint32_t *>(s_ring + rb + ((pos + 8u) & (kIdxRing - 4u)));
        /* ---- request the next chunk of walkers 16j..16j+15 (one bpermute each: requested | want) ---- */
        {
            const bool want = myturn & !done & (requested < vend) & (pos + (kIdxRing - kIdxChunk) >= requested);
            const uint32_t wi = want ? 1u : 0u;
            const uint32_t msg = requested | wi;
            requested += wi * kIdxChunk;
            out128 |= wi * kIdxChunk;
            const uint32_t s1 = 2u * kIdxWpl * grp + wil, s2 = s1 + kIdxWpl;
            const uint32_t m1 = bperm(msg, s1), m2 = bperm(msg, s2);
            const uint32_t o1 = (m1 & ~1u) + piece16, o2 = (m2 & ~1u) + piece16;
            const bool p1 = (m1 & 1u) != 0u, p2 = (m2 & 1u) != 0u;
            fa = p1 ? s1 * kIdxStride + (o1 & (kIdxRing - 1u)) : mytrash;
            fb = p2 ? s2 * kIdxStride + (o2 & (kIdxRing - 1u)) : mytrash;
            /* always two loads per turn (a lane with nothing to fetch re-reads its stream's first 16 bytes): with a
             * fixed number of vector-memory operations per turn the compiler can wait for exactly the chunks it
             * commits (vmcnt(N)); a conditional load made it drain the queue once per round (2.3 us a round) */
            const uint8_t *g1 = src_base + (soff + ((p1 & (o1 < sve)) ? o1 : 0u));
            const uint8_t *g2 = src_base + (soff2 + ((p2 & (o2 < sve2)) ? o2 : 0u));
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(fd) : "v"(g1)); /* "+": the slot keeps its registers round after round (see above) */
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(fe) : "v"(g2));
        }
        /* ---- one hop, branch-free for the two common states (token, match-length extension) ---- */
        {
            /* a walker ends at the first TOKEN position at or behind `stop` (the end of its segment; the stream's end
             * for a block's last walker) */
            const uint32_t lim = state != 0u ? vend : stop;
            const bool live = !done & (pos < lim);
            const bool canread = (pos + 8u <= filled) | (filled >= vend); /* filled <= requested: nothing is read ahead of the requests */
            const uint32_t x = __builtin_amdgcn_alignbyte(w1, w0, pos & 3u);
            const bool go = live & canread;
            /* token */
            const uint32_t ll = (x >> 4) & 15u, e1 = (x >> 8) & 255u, tmn = x & 15u;
            const bool l15 = ll == 15u;
            const uint32_t q2 = pos + 3u + ll + (l15 ? e1 + 1u : 0u);  /* behind the literals and the offset */
            const bool tok = go & (state == 0u);
            /* rare token forms leave the fast path: a literal length that goes on behind its first extension byte, and
             * the stream's last sequence (literals only) */
            const bool rare = tok & ((l15 & (e1 == 255u)) | (q2 > vend));
            const bool plain = tok & !rare;
            /* match-length extension bytes */
            const uint32_t nx = ~x;
            const uint32_t n = nx ? (uint32_t)__builtin_ctz(nx) >> 3 : 4u; /* leading 0xFF bytes */
            const bool extb = go & (state == 2u);
            /* record + advance */
            *(tok ? pbuf + (k & (2u * kIdxLine - 1u)) : ptrash) = (uint16_t)(pos - delta); /* unconditional: a store in a branch costs more than the branch saves */
            k += tok ? 1u : 0u;
            tm = tok ? tmn : tm;
            const uint32_t st_tok = tmn == 15u ? 2u : 0u;
            uint32_t npos = plain ? q2 : pos, nstate = plain ? st_tok : state;
            /* a second token in the same turn when the first one leaves it inside the eight bytes just read: no or
             * up to two literals and a short match (half of the sequences of tuple data) */
            {
                const bool dbl = plain & !l15 & (tmn != 15u) & (ll <= 2u) & (q2 < stop) & (k < kcap);
                const uint32_t x1 = __builtin_amdgcn_alignbyte(w2, w1, pos & 3u); /* bytes 4..7 behind the first token */
                const unsigned long long xx = ((unsigned long long)x1 << 32) | x;
                const uint32_t y = (uint32_t)(xx >> (8u * (3u + ll)));
                const uint32_t llb = (y >> 4) & 15u, e1b = (y >> 8) & 255u, tmb = y & 15u;
                const bool l15b = llb == 15u;
                const uint32_t q2b = q2 + 3u + llb + (l15b ? e1b + 1u : 0u);
                const bool rec2 = dbl & !(l15b & (e1b == 255u));
                *(rec2 ? pbuf + (k & (2u * kIdxLine - 1u)) : ptrash) = (uint16_t)(q2 - delta);
                k += rec2 ? 1u : 0u;
                tm = rec2 ? tmb : tm;
                const bool last2 = rec2 & (q2b > vend); /* the second token is the stream's last sequence */
                const bool adv2 = rec2 & !last2;
                npos = adv2 ? q2b : npos;
                nstate = adv2 ? (tmb == 15u ? 2u : 0u) : nstate;
                done = done | last2;
            }
            npos = extb ? pos + (n == 4u ? 4u : n + 1u) : npos;
            nstate = extb ? (n == 4u ? 2u : 0u) : nstate;
            done = done | !live | (k >= kcap);
            pos = npos;
            state = nstate;
            /* a long match-length run (round 5): the zero gap of a cryo block is ONE match of 100 KB ... 1 MB = 400 ... 4 100
             * extension bytes, four per turn above: 1 025 turns for a 1 MiB block of narrow rows, and the pass over 8 192 such
             * blocks took 1.0-1.6 ms of a 2.6-3.3 ms call (profiles/r05_stream_rot.txt).  When the four bytes just taken were all
             * 255, up to seven more dwords of the ring are looked at: 28 bytes per turn, as much as the ring is fed. */
            {
                const bool longm = extb & (n == 4u) & !done;
                if (wave_any(longm)) {
                    const uint32_t a0 = pos & ~3u; /* the dword that holds the first byte not looked at: what lies before it in there is 255 */
                    uint32_t m = 0;
                    bool run = longm;
#pragma unroll
                    for (uint32_t kq = 0; kq < 7u; kq++) {
                        const uint32_t dq = *reinterpret_cast<const uint32_t *>(s_ring + rb + ((a0 + 4u * kq) & (kIdxRing - 4u)));
                        run = run & (dq == 0xffffffffu) & (a0 + 4u * kq + 4u <= filled);
                        m += run ? 1u : 0u;
                    }
                    const uint32_t far = a0 + 4u * m;
                    pos = (longm & (far > pos)) ? far : pos;
                }
            }
            /* rare: the token forms above, literal-length 255-runs, and jumps over everything requested (a long literal run) */
            const bool slow = rare | (!done & ((state == 1u) | (pos >= requested)));
            if (wave_any(slow)) {
                if (rare) {
                    if (l15 && e1 == 255u) { state = 1u; acc = 15u + 255u; pos += 2u; }
                    else done = true; /* q2 > vend: last sequence */
                } else if (slow && pos < vend) {
                    if (pos >= requested) {
                        /* restart the ring at the chunk of pos; a chunk still on its way lands in a slot that is
                         * rewritten before it is read, and is not counted */
                        requested = filled = pos & ~(kIdxChunk - 1u);
                        outA0 = outA1 = outB0 = outB1 = 0;
                    } else if (state == 1u && live && canread) {
                        if (n == 4u) { acc += 1020u; pos += 4u; if (acc >= vend) done = true; }
                        else {
                            acc += 255u * n + ((x >> (8u * n)) & 255u);
                            const uint32_t q = pos + n + 1u + acc;
                            if (acc >= vend || q + 2u > vend) done = true;
                            else { pos = q + 2u; state = tm == 15u ? 2u : 0u; }
                        }
                    }
                }
            }
        }
    };
    /* Positions go out in whole 32-byte lines of 16, each line stored ONCE, when it is complete (a lane gains at most
     * eight positions per round, so one line per round keeps up; the line being filled meanwhile is the other half of
     * pbuf).  Round 2 stored aligned groups of four, three per round, again and again while they filled: 3.47 GB
     * reached memory for a 0.83 GB index -- the L2 does not hold 65 536 rows' open lines until they are full.  One
     * unconditional pair of stores per round (a lane with nothing to store writes its dummy slot), see the note on
     * the loads.  A macro, not a lambda: captured by a lambda, the packs lived in scratch memory. */
#define IDX_LINE_COPY(ps_, pd_)                                                                               \
    {                                                                                                        \
        const uint4 v0_ = *reinterpret_cast<const uint4 *>(ps_);                                             \
        store16_out<false>(reinterpret_cast<uint8_t *>(pd_), v0_);                           \
        if (kIdxLine == 16u) {                                                                               \
            const uint4 v1_ = *reinterpret_cast<const uint4 *>((ps_) + 8);                                   \
            store16_out<false>(reinterpret_cast<uint8_t *>((pd_) + 8), v1_);                 \
        }                                                                                                    \
    }
#define IDX_PUT()                                                                                            \
    {                                                                                                        \
        const bool st_ = ls < k / kIdxLine;                                                                  \
        const uint16_t *ps_ = pbuf + (ls & 1u) * kIdxLine;                                                   \
        uint16_t *pd_ = st_ ? row + ls * kIdxLine : dummy;                                                   \
        IDX_LINE_COPY(ps_, pd_)                                                                              \
        if (st_) ls++;                                                                                       \
    }
    /* what is left when a walk ends: at most one complete line and the one being filled (stored whole: the entries
     * behind the count are never read, and a row's capacity is a multiple of 16) */
#define IDX_FLUSH()                                                                                          \
    {                                                                                                        \
        IDX_PUT()                                                                                            \
        IDX_PUT()                                                                                            \
        if ((k & (kIdxLine - 1u)) != 0u && ls == k / kIdxLine) {                                             \
            const uint16_t *ps_ = pbuf + (ls & 1u) * kIdxLine;                                               \
            uint16_t *pd_ = row + ls * kIdxLine;                                                             \
            IDX_LINE_COPY(ps_, pd_)                                                                          \
        }                                                                                                    \
    }
#define IDX_TURN(j, n, o, sa, sb) turn(j, fd##n, fe##n, fa##n, fb##n, o, saoff##sa, svend##sa, saoff##sb, svend##sb);
#define IDX_ROUND(a, b, c, d, o0, o1)                           \
    IDX_PUT()                                                   \
    IDX_TURN(0, a, o0, 0, 1) IDX_TURN(1, b, o0, 2, 3) IDX_TURN(2, c, o0, 4, 5) IDX_TURN(3, d, o0, 6, 7)
#define IDX_ROUNDS() IDX_ROUND(0, 1, 2, 3, outA0, outA1) IDX_ROUND(4, 5, 6, 7, outB0, outB1)
    /* (the compiler does not see the assembly loads: nothing it generates behind a walk may meet one still in flight.  The
     * drain names every slot: their registers are dead to the compiler once the loop is left, and without the operands it
     * may hand them to something else in FRONT of the drain -- k_zchain4 of zstd_pipe.hip did, and a late load overwrote
     * an address) */
#define IDX_DRAIN() asm volatile("s_waitcnt vmcnt(0)" : "+v"(fd0), "+v"(fe0), "+v"(fd1), "+v"(fe1), "+v"(fd2), "+v"(fe2), "+v"(fd3), "+v"(fe3), \
                                 "+v"(fd4), "+v"(fe4), "+v"(fd5), "+v"(fe5), "+v"(fd6), "+v"(fe6), "+v"(fd7), "+v"(fe7) : : "memory");
#define IDX_WALK()                                              \
    if (wave_any(!done)) {                                         \
        do {                                                    \
            IDX_ROUNDS()                                        \
        } while (wave_any(!done));                                 \
    }                                                           \
    IDX_DRAIN()                                                 \
    if (walker) IDX_FLUSH()

    /* ---- phase 1: every walker its own segment ---- */
#ifdef CRYO_IDX_PROF
    const unsigned long long t_p0 = __builtin_amdgcn_s_memtime();
#endif
    IDX_WALK()
#ifdef CRYO_IDX_PROF
    const unsigned long long t_p1 = __builtin_amdgcn_s_memtime();
    uint32_t prof_steps = 0;
#endif

    uint32_t d_ext = 0, d_skip = 0, d_cnt = walker ? k : 0u; /* this segment's descriptor */
    if (logS != 0u) {
        /* ---- phase 2 (round 5 form): which walkers lie on the true chain, and each of those takes the chain of the one before it
         * into its own segment until it stands on one of its own records.
         *
         * Rounds 3-4 let walker s walk on into segment s+1 and failed the block when its chain jumped over that segment -- which
         * is what the chain of a cryo block of narrow rows does: the zero gap is ONE sequence whose match-length bytes alone
         * (4 100 of them for 1 MiB) span several 1 KiB segments, every such block took the one-walker redo, and the pass was
         * half of the call on those shapes (profiles/r05_index_run255.txt).  Now: a walker's chain leaves its segment at
         * `pos` = the first token behind it, in segment nseg > s.  Walker 0 is true; if walker u is true (from its meeting
         * point on), so is where it ends, hence walker nseg(u) is the next one on the chain and the segments in between hold
         * no token at all.  One sweep over the wave's lanes in order (a chain only goes forward) marks the walkers on the chain
         * and their predecessors; the others describe nothing. ---- */
        const bool lastw = walker && sw + 1u == seff;
        /* an inner walker must have ended on a token behind its segment (not: records overflowed, stream cut, 255-run) */
        const bool endok = walker && (lastw || (pos >= stop && pos < vend && state == 0u && k < kcap));
        uint32_t nlane = 64u; /* the lane of the walker whose segment this walker's chain enters (64: none) */
        if (endok && !lastw) {
            const uint32_t t = (pos - delta) / seglen; /* pos < vend: t < seff */
            nlane = lane - sw + (t < seff ? t : seff - 1u);
        }
        unsigned long long lm = wave_ballot(walker && sw == 0u);
        uint32_t pred = 0;
        for (uint32_t u = 0; u < 64u; u++) {
            if ((lm >> u) & 1ull) {
                const uint32_t tu = lane_get(nlane, u);
                if (tu < 64u) {
                    lm |= 1ull << tu;
                    pred = lane == tu ? u : pred;
                }
            }
        }
        const bool live = walker && ((lm >> lane) & 1ull) != 0ull;
        bool fail = live && !endok;
        __threadfence(); /* records are read back from memory */
        const uint32_t pin = bperm(pos, pred); /* where the chain enters this segment */
        const uint16_t *rec = row;             /* this walker's records: the first one is its guessed start */
        uint16_t *my_ext = row - ext;          /* ... and the segment's extension in front of them */
        const uint8_t *sb = src_base + aoff;
        uint32_t p = pin, r = gstart, j = 0, L = 0;
        bool merging = live && sw != 0u && !fail;
        if (merging && (k == 0u || !(p >= gstart && p < stop))) { fail = true; merging = false; } /* (a walker on the chain has records) */
        if (merging) r = gstart + (((uint32_t)rec[0] + delta - gstart) & 0xffffu); /* its first record: at or behind the guessed start (state 2) */
        while (wave_any(merging)) {
#ifdef CRYO_IDX_PROF
            prof_steps++;
#endif
            if (merging) {
                if (p == r) merging = false;
                else if (p < r) {
                    if (L >= ext || p >= stop || p >= vend) { fail = true; merging = false; }
                    else {
                        my_ext[L++] = (uint16_t)(p - delta);
                        p = lz4_next_token_direct(sb, vend, p);
                    }
                } else {
                    /* p is ahead of record j: step through the records, sixteen per trip to memory (one record per trip made
                     * literal-heavy streams, whose false chains record a position every ~7 bytes of a long literal run, pay
                     * 500-1000 dependent trips here: 3.2 ms for 8 192 blocks) */
                    if (j >= k) { fail = true; merging = false; } /* already behind the walker's end: the chains did not meet */
                    else {
                        uint4 va, vb;
                        __builtin_memcpy(&va, rec + j + 1u, 16); /* records j+1 .. j+16; beyond the count: never used */
                        __builtin_memcpy(&vb, rec + j + 9u, 16);
                        const uint32_t w[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
                        for (uint32_t t = 0; t < 16u; t++) {
                            const uint32_t e16 = (w[t >> 1] >> (16u * (t & 1u))) & 0xffffu;
                            const bool step = (r < p) & (j + 1u < k);
                            r += step ? ((e16 + delta - r) & 0xffffu) : 0u; /* records are increasing, less than 64 KiB apart */
                            j += step ? 1u : 0u;
                        }
                        if ((r < p) & (j + 1u == k)) { j = k; r = pos; } /* behind its last record the walker stands on its end */
                    }
                }
            }
        }
        /* The skip count travels in 16 bits: a hand-over deeper than 65 535 records into the segment (two walkers on a 1 MiB
         * block of dense sequences can get there) counts as a boundary that did not meet, and phase 3 walks the block with one
         * walker */
        fail = fail || (live && j > 0xffffu);
        const uint32_t left = L | (j << 16);
        const unsigned long long fm = wave_ballot(fail);
        const unsigned long long gmask = (logS >= 6u ? ~0ull : ((1ull << S) - 1ull)) << (lane & ~(S - 1u));
        const bool gfail = (fm & gmask) != 0ull;
        if (!gfail && walker && !live) d_cnt = 0; /* not on the chain: the segment holds no token */
        if (!gfail) {
            if (live && sw != 0u) { d_ext = left & 0xffffu; d_skip = left >> 16; d_cnt = k - d_skip; }
        } else {
            /* ---- phase 3: the chains of this block did not meet: its first lane walks all of it ---- */
            d_cnt = 0;
        }
        if (wave_any(gfail)) {
            const bool redo = gfail && walker && sw == 0u;
            pos = delta;
            requested = filled = 0;
            outA0 = outA1 = outB0 = outB1 = 0; /* a chunk still on its way is not counted */
            state = 0; acc = 0; tm = 0; k = 0; ls = 0;
            stop = vend;
            kcap = S * cap_s - ext;
            done = !redo || vend == delta;
            {
                const bool walker = redo; /* who flushes */
                IDX_WALK()
            }
            if (redo) d_cnt = k;
        }
    }
#undef IDX_WALK
#undef IDX_DRAIN
#undef IDX_ROUNDS
#undef IDX_ROUND
#undef IDX_TURN
#undef IDX_FLUSH
#undef IDX_PUT
    if (owner) seg[gl] = make_uint2(d_ext | (d_skip << 16), d_cnt);
#ifdef CRYO_IDX_PROF
    if ((blockIdx.x & 63u) == 0u && lane == 0u)
        printf("[index] wave %u: walk %llu ticks, hand-over and the rest %llu ticks, %u merge steps\n", blockIdx.x, t_p1 - t_p0, __builtin_amdgcn_s_memtime() - t_p1, prof_steps);
#endif
}

/* ---------------------------------------------------------------------------------------------
 * Few blocks per call (round 5): up to 1 024 walkers per block, reading the stream straight from memory.
 *
 * k_lz4_index above is built for full batches: 65 536 streams come from HBM, so every walker stages its stream through an
 * LDS ring, and the rings cap a block at 64 walkers (one wave; LDS: 39 KB per wave).  For the call shapes of the unmodified
 * reference -- ONE block per call, 16 cache slots (pg_cryogen.c:726, cache.c:17,178) -- that is the wrong machine: the chip is
 * empty, a block's stream sits in the caches after its first touch, and the walk of 64 walkers x 16 KiB is the call (0.30 of
 * the 0.41 ms of one 1 MiB block, profiles/r03_lz4_few_blocks.txt).  Here a lane walks a segment of 1-2 KiB with direct
 * loads (a hop = a trip to the L2), records go straight to its row; a block's walkers are S / 64 independent waves spread
 * over the CUs (as ONE workgroup of 1 024 lanes they shared a CU's address path and instruction issue: 0.16 ms per 1 MiB block
 * against 0.27 with the rings; profiles/r05_lz4_few_blocks.txt), and what k_lz4_index exchanges between neighbouring lanes
 * with DPP shifts goes through memory and a kernel boundary:
 *   k_lz4_few_walk  every walker its own segment from a guessed start; its record count and the position it ends on -> kk, ee
 *   k_lz4_few_join  walker s takes the walk of its LEFT neighbour from where that one ended into its own segment until it
 *                   stands on one of its own records: the positions on the way are segment s's extension, the records in
 *                   front of the meeting point are skipped -- the same two pieces per segment, the same descriptors as
 *                   k_lz4_index writes (lz4_lat.hip reads them).
 * A boundary whose chains do not meet, or a segment with more tokens than its row holds, marks the block (failed[blk]): the
 * caller's other decoder takes it.
 * --------------------------------------------------------------------------------------------- */
struct FewGeom {
    uint32_t cs, seff, seglen;
    bool indexed;
};
__device__ inline FewGeom few_geom(const uint32_t cs, const uint32_t logS, const uint32_t block_size)
{
    FewGeom g;
    g.cs = cs;
    const uint32_t kib = cs >> 10; /* segments exactly as k_lz4_index cuts them: at least 1 KiB each */
    const uint32_t lg = kib ? 31u - (uint32_t)__builtin_clz(kib) : 0u;
    const bool cs_small = lz4_index_one_walker(cs);
    const uint32_t ls_ = cs_small ? 0u : (lg < logS ? lg : logS);
    g.seff = 1u << ls_;
    g.seglen = (cs + g.seff - 1u) >> ls_;
    g.indexed = cs != 0u && !lz4_literal_heavy(cs, block_size);
    return g;
}

__global__ void __launch_bounds__(64)
k_lz4_few_walk(const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off, const uint32_t *__restrict__ src_size,
               uint16_t *__restrict__ tbl, const uint32_t logS, const uint32_t cap_main, const uint32_t ext, const uint32_t cap,
               uint32_t *__restrict__ kk, uint32_t *__restrict__ ee, uint32_t *__restrict__ failed, const uint32_t block_size)
{
    const uint64_t blk = blockIdx.y;
    const uint32_t sw = blockIdx.x * 64u + threadIdx.x, cap_s = cap_main + ext;
    const FewGeom g = few_geom(src_size[blk], logS, block_size);
    const uint8_t *sb = src_base + src_off[blk];
    if (sw == 0u) failed[blk] = 0u;
    const bool walker = g.indexed && sw < g.seff;
    const uint32_t gstart = sw * g.seglen;
    const uint32_t stop = (walker && sw + 1u < g.seff) ? (sw + 1u) * g.seglen : g.cs;
    uint16_t *row = tbl + blk * cap + sw * cap_s + ext;
    uint32_t p = gstart, k = 0;
    if (walker) {
        /* every line of the segment is asked for at once (the walk would take the misses one after the other) */
        uint32_t warm = 0;
        for (uint32_t o = gstart & ~127u; o < stop; o += 128u) warm += sb[o < g.cs ? o : 0u];
        asm volatile("" ::"v"(warm));
        /* a guessed start inside a run of 255s (the match-length bytes of the zero gap, several segments long) moves behind the
         * run's end: a true token there.  (Read as a token, a 255 is a literal length of megabytes: out of the stream.) */
        if (sw != 0u && sb[p] == 255u) p = lz4_skip_255(sb, g.cs, p + 1u, nullptr);
        const uint32_t kcap = g.seff == 1u ? (cap_s << logS) - ext : cap_main; /* a block's only walker has the whole row */
        while (p < stop) {
            if (k >= kcap) { p = 0xfffffffeu; break; } /* more tokens than the row holds: the join sees a walker that did not end on a token */
            uint32_t x;
            __builtin_memcpy(&x, sb + p, 4);         /* the hop's load first, the record's store behind it (lz4_next_token_word) */
            asm volatile("" ::: "memory");
            row[k++] = (uint16_t)p;
            p = lz4_next_token_word(sb, g.cs, p, x); /* >= cs behind the stream's last sequence */
        }
    }
    kk[(blk << logS) + sw] = walker ? k : 0u;
    ee[(blk << logS) + sw] = p;
}

/* which walkers lie on the true chain (k_lz4_index's phase 2 has the reasoning): one wave per block sweeps its up to 1 024
 * walkers in order, 64 at a time; lp[walker] = 0x80000000 | predecessor for the walkers on the chain, 0 for the others */
__global__ void __launch_bounds__(64)
k_lz4_few_path(const uint32_t *__restrict__ src_size, const uint32_t logS, const uint32_t *__restrict__ ee, uint32_t *__restrict__ lp,
               const uint32_t block_size)
{
    __shared__ uint32_t s_live[32];
    __shared__ uint32_t s_pred[1024];
    const uint64_t blk = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const FewGeom g = few_geom(src_size[blk], logS, block_size);
    if (lane < 32u) s_live[lane] = (lane == 0u && g.indexed) ? 1u : 0u;
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const uint32_t S = 1u << logS;
    for (uint32_t u0 = 0; u0 < S; u0 += 64u) {
        const uint32_t sw = u0 + lane;
        const bool walker = g.indexed && sw < g.seff;
        uint32_t nseg = 0xffffu; /* the segment this walker's chain enters (none: it is the last one, or it did not end on a token) */
        if (walker && sw + 1u < g.seff) {
            const uint32_t e = ee[(blk << logS) + sw], stop = (sw + 1u) * g.seglen;
            if (e >= stop && e < g.cs) { const uint32_t t = e / g.seglen; nseg = t < g.seff ? t : g.seff - 1u; }
        }
        unsigned long long lm = (unsigned long long)uni(s_live[u0 >> 5]) | ((unsigned long long)uni(s_live[(u0 >> 5) + 1u]) << 32);
        uint32_t pred = s_pred[sw]; /* written by an earlier group of 64, if this walker is on the chain */
        for (uint32_t u = 0; u < 64u; u++) {
            if ((lm >> u) & 1ull) {
                const uint32_t tu = lane_get(nseg, u);
                if (tu < g.seff) {
                    if (tu - u0 < 64u) {
                        lm |= 1ull << (tu - u0);
                        pred = lane == tu - u0 ? u0 + u : pred;
                    } else if (lane == 0u) {
                        s_live[tu >> 5] |= 1u << (tu & 31u);
                        s_pred[tu] = u0 + u;
                    }
                }
            }
        }
        const bool live = walker && ((lm >> lane) & 1ull) != 0ull;
        if (sw < S) lp[(blk << logS) + sw] = live ? (0x80000000u | pred) : 0u;
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    }
}

__global__ void __launch_bounds__(64)
k_lz4_few_join(const uint8_t *__restrict__ src_base, const uint64_t *__restrict__ src_off, const uint32_t *__restrict__ src_size,
               uint16_t *__restrict__ tbl, const uint32_t logS, const uint32_t cap_main, const uint32_t ext, const uint32_t cap,
               const uint32_t *__restrict__ kk, const uint32_t *__restrict__ ee, const uint32_t *__restrict__ lp,
               uint32_t *__restrict__ failed, uint2 *__restrict__ seg, const uint32_t block_size)
{
    const uint64_t blk = blockIdx.y;
    const uint32_t sw = blockIdx.x * 64u + threadIdx.x, cap_s = cap_main + ext;
    const FewGeom g = few_geom(src_size[blk], logS, block_size);
    const uint8_t *sb = src_base + src_off[blk];
    const bool walker = g.indexed && sw < g.seff;
    const uint32_t k = kk[(blk << logS) + sw], e = ee[(blk << logS) + sw];
    const uint32_t gstart = sw * g.seglen;
    const uint32_t stop = (walker && sw + 1u < g.seff) ? (sw + 1u) * g.seglen : g.cs;
    /* only the walkers on the true chain describe anything (k_lz4_few_path); such a walker's own walk must have ended on a
     * token behind its segment (an inner one) or behind the stream (the last) */
    const uint32_t lpv = lp[(blk << logS) + sw];
    const bool live = walker && (lpv >> 31) != 0u;
    bool fail = live && (e == 0xfffffffeu || (sw + 1u < g.seff && !(e >= stop && e < g.cs)));
    uint32_t L = 0, j = 0;
    if (live && sw != 0u && !fail) {
        const uint16_t *rec = tbl + blk * cap + sw * cap_s + ext; /* this walker's records: the first one is its guessed start */
        uint16_t *my_ext = tbl + blk * cap + sw * cap_s;          /* ... and the extension in front of them                 */
        uint32_t pp = ee[(blk << logS) + (lpv & 0x7fffffffu)], r = gstart; /* where the chain enters this segment; record j of this walker */
        if (k == 0u || !(pp >= gstart && pp < stop)) fail = true;           /* (a walker on the chain has records) */
        else r = gstart + (((uint32_t)rec[0] - gstart) & 0xffffu);           /* its first record: at or behind the guessed start */
        while (!fail) {
            if (pp == r) break;
            if (pp < r) {
                if (L >= ext || pp >= stop || pp >= g.cs) { fail = true; break; }
                uint32_t x;
                __builtin_memcpy(&x, sb + pp, 4);
                asm volatile("" ::: "memory");
                my_ext[L++] = (uint16_t)pp;
                pp = lz4_next_token_word(sb, g.cs, pp, x);
            } else {
                if (j >= k) { fail = true; break; }          /* already behind e: the chains did not meet */
                if (j + 1u < k) { j++; r += ((uint32_t)rec[j] - r) & 0xffffu; } /* records are increasing, less than 64 KiB apart */
                else { j = k; r = e; }                        /* behind its last record the walker stands on e */
            }
        }
        fail = fail || j > 0xffffu;
    }
    if (fail) failed[blk] = 1u;
    seg[(blk << logS) + sw] = (live && !fail) ? make_uint2(L | (j << 16), k - j) : make_uint2(0u, 0u);
}

/* ---- layout and launcher ---- */
Lz4IndexLayout lz4_index_layout(uint64_t n_blocks, uint32_t block_size, uint32_t walkers)
{
    Lz4IndexLayout L;
    uint32_t lg = 0;
    while ((2u << lg) <= walkers && lg < 6u) lg++;
    L.logS = lg;
    const uint32_t S = 1u << lg;
    /* a block of B bytes holds at most B/8 + 64 sequences the decoder's batches can use (the rest is decoded without
     * hints); a segment is 1/S of the compressed bytes, not of the sequences: a quarter more */
    L.ext = S > 1u ? 512u : 0u;
    const uint32_t per = (block_size / 8u + 64u + S - 1u) / S;
    L.cap_main = S > 1u ? ((per + per / 4u + 63u) & ~63u) : per;
    /* rows 2 KiB-granular: the pass writes 64 rows at once and is sensitive to the row stride (33 024-byte rows always
     * ran 17 % slower, 34 816 never: profiles/scripts/r02_cap.sh) */
    L.cap = (S * (L.cap_main + L.ext) + 1023u) & ~1023u;
    if (S == 1u) L.cap_main = L.cap;
    const size_t tbl_bytes = (size_t)n_blocks * L.cap * 2u;
    L.dummy_off = tbl_bytes;
    L.seg_off = (tbl_bytes + 64u * 32u + 255u) & ~(size_t)255u;
    const uint64_t lanes = ((n_blocks << lg) + 63u) & ~(uint64_t)63u;
    L.bytes = L.seg_off + (size_t)lanes * sizeof(uint2) + 256u;
    return L;
}

/* the few-blocks form: S = up to 1 024 walkers per block (segments of 1-2 KiB); a segment's row holds what 2 KiB of stream can
 * hold at three bytes per sequence, its extension 128 entries */
Lz4IndexLayout lz4_index_layout_few(uint64_t n_blocks, uint32_t block_size)
{
    Lz4IndexLayout L;
    uint32_t lg = 6;
    while (lg < 10u && (block_size >> (lg + 1u)) >= 1024u) lg++; /* no more walkers than KiB of block */
    L.logS = lg;
    const uint32_t S = 1u << lg;
    L.ext = 128u;
    L.cap_main = 704u; /* >= 2047 / 3 + 1, a multiple of 64 */
    L.cap = S * (L.cap_main + L.ext);
    const size_t tbl_bytes = (size_t)n_blocks * L.cap * 2u;
    L.dummy_off = tbl_bytes;
    L.seg_off = (tbl_bytes + 64u * 32u + 255u) & ~(size_t)255u;
    /* behind the descriptors: record counts and end positions of the walkers (k_lz4_few_walk -> k_lz4_few_join), the blocks' flags */
    L.bytes = L.seg_off + ((size_t)n_blocks << lg) * (sizeof(uint2) + 12u) + n_blocks * 4u + 256u;
    return L;
}
/* where the few-blocks index keeps "this block has no index" (the walkers' chains did not meet): one word per block */
const uint32_t *lz4_index_few_failed(const void *d_workspace, const Lz4IndexLayout &L, uint64_t n_blocks)
{
    return reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(d_workspace) + L.seg_off + ((size_t)n_blocks << L.logS) * (sizeof(uint2) + 12u));
}

hipError_t launch_lz4_index_few(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off, const uint32_t *d_src_size,
                                uint64_t n_blocks, uint32_t block_size, void *d_workspace, const Lz4IndexLayout &L)
{
    if (n_blocks == 0) return hipSuccess;
    if (n_blocks > 65535u || L.logS < 6u || L.logS > 10u) return hipErrorInvalidValue;
    uint8_t *ws = static_cast<uint8_t *>(d_workspace);
    uint16_t *tbl = reinterpret_cast<uint16_t *>(ws);
    uint2 *seg = reinterpret_cast<uint2 *>(ws + L.seg_off);
    uint32_t *kk = reinterpret_cast<uint32_t *>(ws + L.seg_off + ((size_t)n_blocks << L.logS) * sizeof(uint2));
    uint32_t *ee = kk + ((size_t)n_blocks << L.logS);
    uint32_t *lp = ee + ((size_t)n_blocks << L.logS);
    uint32_t *failed = lp + ((size_t)n_blocks << L.logS);
    const dim3 grid(1u << (L.logS - 6u), (uint32_t)n_blocks);
    hipLaunchKernelGGL(k_lz4_few_walk, grid, dim3(64), 0, s, d_src, d_src_off, d_src_size, tbl, L.logS, L.cap_main, L.ext, L.cap, kk, ee, failed, block_size);
    hipLaunchKernelGGL(k_lz4_few_path, dim3((uint32_t)n_blocks), dim3(64), 0, s, d_src_size, L.logS, ee, lp, block_size);
    hipLaunchKernelGGL(k_lz4_few_join, grid, dim3(64), 0, s, d_src, d_src_off, d_src_size, tbl, L.logS, L.cap_main, L.ext, L.cap, kk, ee, lp, failed, seg, block_size);
    return hipGetLastError();
}

hipError_t launch_lz4_index(hipStream_t s, const uint8_t *d_src, const uint64_t *d_src_off, const uint32_t *d_src_size,
                            uint64_t n_blocks, uint32_t block_size, void *d_workspace, const Lz4IndexLayout &L)
{
    if (n_blocks == 0) return hipSuccess;
    const uint64_t grid = ((n_blocks << L.logS) + kIdxLanes - 1) / kIdxLanes;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    uint8_t *ws = static_cast<uint8_t *>(d_workspace);
    hipLaunchKernelGGL(k_lz4_index, dim3((uint32_t)grid), dim3(64), 0, s, d_src, d_src_off, d_src_size, n_blocks,
                       reinterpret_cast<uint16_t *>(ws), L.logS, L.cap_main, L.ext, L.cap, reinterpret_cast<uint2 *>(ws + L.seg_off),
                       reinterpret_cast<uint16_t *>(ws + L.dummy_off), block_size);
    return hipGetLastError();
}

} // namespace cryo
