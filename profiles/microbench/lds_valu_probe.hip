// Microbenchmarks that shaped the round-2 LZ4 copy engine: VALU issue rate per SIMD vs waves per SIMD,
// and the cost of scattered / unaligned wide LDS accesses (ds_read_b64 / ds_write_b64 at byte addresses).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_valu_probe lds_valu_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;

// ---------------- VALU issue ----------------
template <int KIND>
__global__ void k_valu(uint32_t *out, unsigned long long *cyc)
{
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 9, a5 = a0 ^ 77, a6 = a0 * 11, a7 = a0 + 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; i++) {
        if (KIND == 0) {
            asm volatile("v_add_u32 %0, %0, %1\nv_add_u32 %1, %1, %2\nv_add_u32 %2, %2, %3\nv_add_u32 %3, %3, %4\n"
                         "v_add_u32 %4, %4, %5\nv_add_u32 %5, %5, %6\nv_add_u32 %6, %6, %7\nv_add_u32 %7, %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 1) {
            asm volatile("v_perm_b32 %0, %0, %1, %2\nv_perm_b32 %1, %1, %2, %3\nv_perm_b32 %2, %2, %3, %4\nv_perm_b32 %3, %3, %4, %5\n"
                         "v_perm_b32 %4, %4, %5, %6\nv_perm_b32 %5, %5, %6, %7\nv_perm_b32 %6, %6, %7, %0\nv_perm_b32 %7, %7, %0, %1\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 2) {
            asm volatile("v_lshl_add_u32 %0, %0, 1, %1\nv_and_or_b32 %1, %1, %2, %3\nv_bfe_u32 %2, %2, 3, 9\nv_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_lshlrev_b32 %4, 1, %5\nv_xor_b32 %5, %5, %6\nv_min_u32 %6, %6, %7\nv_sub_u32 %7, %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
        } else if (KIND == 3) { /* DPP moves */
            asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %4 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %6 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 4) { /* 64-bit shifts */
            unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3;
            asm volatile("v_lshlrev_b64 %0, 3, %1\nv_lshrrev_b64 %1, 5, %2\nv_lshlrev_b64 %2, 3, %3\nv_lshrrev_b64 %3, 5, %0\n"
                         "v_lshlrev_b64 %0, 3, %1\nv_lshrrev_b64 %1, 5, %2\nv_lshlrev_b64 %2, 3, %3\nv_lshrrev_b64 %3, 5, %0\n"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
            a0 = (uint32_t)b0; a1 = (uint32_t)b1; a2 = (uint32_t)b2; a3 = (uint32_t)b3;
        } else if (KIND == 5) { /* SALU mix */
            uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4;
            asm volatile("s_add_u32 %0, %0, %1\ns_lshl_b32 %1, %1, 1\ns_and_b32 %2, %2, %3\ns_xor_b32 %3, %3, %0\n"
                         "s_add_u32 %0, %0, %1\ns_lshl_b32 %1, %1, 1\ns_and_b32 %2, %2, %3\ns_xor_b32 %3, %3, %0\n"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
            a0 += s0;
        } else if (KIND == 6) { /* half VALU half SALU interleaved */
            uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4;
            asm volatile("v_add_u32 %0, %0, %1\ns_add_u32 %4, %4, %5\nv_add_u32 %1, %1, %2\ns_lshl_b32 %5, %5, 1\n"
                         "v_add_u32 %2, %2, %3\ns_and_b32 %6, %6, %7\nv_add_u32 %3, %3, %0\ns_xor_b32 %7, %7, %4\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
            a0 += s0;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// ---------------- LDS scattered ----------------
// per wave region of RG bytes; lane address = base[lane] advanced by `step` per op, masked into the region
constexpr int RG = 8192;
enum { RD64, RD32, RD128, WR64, WR32, WR16, WR8, RD8, RDWR64, RD16 };
template <int OP>
__global__ void k_lds(const uint32_t *addr0, uint32_t step, uint32_t active, uint32_t *out, unsigned long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t rbase = wid * (RG + 64);
    for (uint32_t i = lane * 4; i < RG + 64; i += 256) *reinterpret_cast<uint32_t *>(smem + rbase + i) = i * 2654435761u;
    __syncthreads();
    uint32_t a = addr0[lane];
    uint32_t acc = 0;
    unsigned long long v = lane, v2 = 0;
    const bool on = lane < active;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (on) {
        for (int i = 0; i < ITERS / 8; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t p = rbase + (a & (RG - 1));
                if (OP == RD64) { asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(p)); }
                if (OP == RD32) { uint32_t x; asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(p)); v = x; }
                if (OP == RD16) { uint32_t x; asm volatile("ds_read_u16 %0, %1" : "=v"(x) : "v"(p)); v = x; }
                if (OP == RD8) { uint32_t x; asm volatile("ds_read_u8 %0, %1" : "=v"(x) : "v"(p)); v = x; }
                if (OP == RD128) { uint4 x; asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(p)); v = x.x; }
                if (OP == WR64) { asm volatile("ds_write_b64 %0, %1" : : "v"(p), "v"(v)); }
                if (OP == WR32) { asm volatile("ds_write_b32 %0, %1" : : "v"(p), "v"((uint32_t)v)); }
                if (OP == WR16) { asm volatile("ds_write_b16 %0, %1" : : "v"(p), "v"((uint32_t)v)); }
                if (OP == WR8) { asm volatile("ds_write_b8 %0, %1" : : "v"(p), "v"((uint32_t)v)); }
                if (OP == RDWR64) { asm volatile("ds_read_b64 %0, %1" : "=v"(v2) : "v"(p)); asm volatile("ds_write_b64 %0, %1" : : "v"(rbase + ((a + 4099u) & (RG - 1))), "v"(v)); }
                a += step;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc += (uint32_t)v + (uint32_t)v2;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wid] = t1 - t0;
}

// correctness of unaligned wide LDS access
__global__ void k_unaligned_check(uint32_t *bad)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[2048];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 2048; i += 64) s[i] = (uint8_t)(i * 31 + 7);
    __syncthreads();
    uint32_t nb = 0;
    for (uint32_t sh = 0; sh < 16; sh++) {
        const uint32_t p = lane * 19 + sh;
        unsigned long long v; uint32_t w, h; uint4 q;
        asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(s + p) & 0xffffu) : "memory");
        asm volatile("ds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(w) : "v"((uint32_t)(uintptr_t)(s + p + 3) & 0xffffu) : "memory");
        asm volatile("ds_read_u16 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(h) : "v"((uint32_t)(uintptr_t)(s + p + 5) & 0xffffu) : "memory");
        asm volatile("ds_read_b128 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(q) : "v"((uint32_t)(uintptr_t)(s + p + 1) & 0xffffu) : "memory");
        for (int k = 0; k < 8; k++) if (((v >> (8 * k)) & 0xff) != (uint8_t)((p + k) * 31 + 7)) nb |= 1;
        for (int k = 0; k < 4; k++) if (((w >> (8 * k)) & 0xff) != (uint8_t)((p + 3 + k) * 31 + 7)) nb |= 2;
        for (int k = 0; k < 2; k++) if (((h >> (8 * k)) & 0xff) != (uint8_t)((p + 5 + k) * 31 + 7)) nb |= 4;
        const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
        for (int k = 0; k < 16; k++) if (((qq[k / 4] >> (8 * (k & 3))) & 0xff) != (uint8_t)((p + 1 + k) * 31 + 7)) nb |= 8;
    }
    __syncthreads();
    // unaligned writes: lane l writes 8 bytes at l*24 + (l % 7), 4 bytes at 1600 + l*6+1, 2 bytes at ...
    for (uint32_t i = lane; i < 2048; i += 64) s[i] = 0;
    __syncthreads();
    {
        const uint32_t p = lane * 24 + (lane % 7);
        const unsigned long long v = 0x0807060504030201ull + lane;
        asm volatile("ds_write_b64 %0, %1\ns_waitcnt lgkmcnt(0)" : : "v"((uint32_t)(uintptr_t)(s + p) & 0xffffu), "v"(v) : "memory");
        const uint32_t p2 = 1600 + lane * 6 + 1;
        asm volatile("ds_write_b32 %0, %1\ns_waitcnt lgkmcnt(0)" : : "v"((uint32_t)(uintptr_t)(s + p2) & 0xffffu), "v"(0xa1b2c3d4u + lane) : "memory");
    }
    __syncthreads();
    {
        const uint32_t p = lane * 24 + (lane % 7);
        const unsigned long long v = 0x0807060504030201ull + lane;
        for (int k = 0; k < 8; k++) if (s[p + k] != (uint8_t)(v >> (8 * k))) nb |= 16;
        const uint32_t p2 = 1600 + lane * 6 + 1, w = 0xa1b2c3d4u + lane;
        for (int k = 0; k < 4; k++) if (s[p2 + k] != (uint8_t)(w >> (8 * k))) nb |= 32;
    }
    if (nb) atomicOr(bad, nb);
}

// unaligned 8-byte global loads at random offsets inside a 64 KiB window per wave (far-match read-back)
__global__ void k_gfar(const uint8_t *buf, uint64_t stride, const uint32_t *addr0, uint32_t *out, unsigned long long *cyc)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wv = (uint64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const uint8_t *base = buf + wv * stride;
    uint32_t a = addr0[lane] * 8u + lane;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 256; i++) {
        unsigned long long v;
        __builtin_memcpy(&v, base + (a & 0xffffu), 8);
        acc += (uint32_t)v;
        a = a * 1664525u + 1013904223u;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[wv * 64 + lane] = acc;
    if (lane == 0) cyc[wv] = t1 - t0;
}

static double median(std::vector<unsigned long long> &v) { std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; }

int main()
{
    int ncu = 256;
    uint32_t *d_out; unsigned long long *d_cyc; uint32_t *d_addr;
    CK(hipMalloc(&d_out, 256 * 16 * 1024 * 4)); CK(hipMalloc(&d_cyc, 65536 * 8)); CK(hipMalloc(&d_addr, 256));
    std::vector<unsigned long long> h(65536);
    // ---- VALU
    const char *vn[] = {"v_add_u32", "v_perm_b32", "mixed int (lshl_add,and_or,bfe,cndmask,shl,xor,min,sub)", "v_mov_dpp row_shr", "v_lsh*_b64", "SALU only", "VALU+SALU 1:1"};
    auto run_valu = [&](int kind, int wps) {
        const int threads = 64 * 4 * wps; // one workgroup per CU with 4*wps waves
        dim3 g(ncu), b(threads);
        for (int rep = 0; rep < 2; rep++) {
            switch (kind) {
            case 0: hipLaunchKernelGGL(k_valu<0>, g, b, 0, 0, d_out, d_cyc); break;
            case 1: hipLaunchKernelGGL(k_valu<1>, g, b, 0, 0, d_out, d_cyc); break;
            case 2: hipLaunchKernelGGL(k_valu<2>, g, b, 0, 0, d_out, d_cyc); break;
            case 3: hipLaunchKernelGGL(k_valu<3>, g, b, 0, 0, d_out, d_cyc); break;
            case 4: hipLaunchKernelGGL(k_valu<4>, g, b, 0, 0, d_out, d_cyc); break;
            case 5: hipLaunchKernelGGL(k_valu<5>, g, b, 0, 0, d_out, d_cyc); break;
            case 6: hipLaunchKernelGGL(k_valu<6>, g, b, 0, 0, d_out, d_cyc); break;
            }
            CK(hipDeviceSynchronize());
        }
        const int nw = ncu * 4 * wps;
        CK(hipMemcpy(h.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> v(h.begin(), h.begin() + nw);
        const double c = median(v);
        printf("VALU %-62s waves/SIMD %d: %.2f cycles per wave-instr per wave -> %.2f cycles per instr per SIMD\n", vn[kind], wps,
               c / (ITERS * 8.0), c / (ITERS * 8.0) / wps);
    };
    for (int kind = 0; kind < 7; kind++) for (int wps : {1, 2, 4}) run_valu(kind, wps);

    // ---- unaligned correctness
    uint32_t *d_bad; CK(hipMalloc(&d_bad, 4)); CK(hipMemset(d_bad, 0, 4));
    hipLaunchKernelGGL(k_unaligned_check, dim3(1), dim3(64), 0, 0, d_bad);
    uint32_t bad; CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("unaligned LDS access check: bad mask = 0x%x (0 = all ds_read_b64/b32/u16/b128 and ds_write_b64/b32 at odd byte addresses are exact)\n", bad);

    // ---- LDS scattered
    const char *on[] = {"ds_read_b64", "ds_read_b32", "ds_read_b128", "ds_write_b64", "ds_write_b32", "ds_write_b16", "ds_write_b8", "ds_read_u8", "ds_read_b64+ds_write_b64", "ds_read_u16"};
    struct Pat { const char *name; int kind; uint32_t p0, p1; };
    // kind 0: lane*p0 + p1 (regular stride), kind 1: random aligned to p0 plus p1, kind 2: lane*p0 + random jitter < p1
    const Pat pats[] = {
        {"consecutive (lane*size)", 9, 0, 0},
        {"stride 20 B, unaligned (+1)", 0, 20, 1},
        {"stride 21 B + jitter<8", 2, 21, 8},
        {"stride 16 B + jitter<8", 2, 16, 8},
        {"random, 8-aligned", 1, 8, 0},
        {"random, 4-aligned", 1, 4, 0},
        {"random byte address", 1, 1, 0},
    };
    const int sizes[] = {8, 4, 16, 8, 4, 2, 1, 1, 8, 2};
    auto run_lds = [&](int op, const Pat &pt, uint32_t step, uint32_t active, int wpc) {
        uint32_t ha[64];
        srand(1234);
        for (int l = 0; l < 64; l++) {
            uint32_t a;
            if (pt.kind == 9) a = l * sizes[op];
            else if (pt.kind == 0) a = l * pt.p0 + pt.p1;
            else if (pt.kind == 1) a = ((uint32_t)rand() % (RG / pt.p0)) * pt.p0 + pt.p1;
            else a = l * pt.p0 + (uint32_t)rand() % pt.p1;
            ha[l] = a;
        }
        CK(hipMemcpy(d_addr, ha, 256, hipMemcpyHostToDevice));
        dim3 g(ncu), b(64 * wpc);
        const size_t sm = (size_t)wpc * (RG + 64);
        for (int rep = 0; rep < 2; rep++) {
            switch (op) {
            case RD64: hipLaunchKernelGGL(k_lds<RD64>, g, b, sm, 0, d_addr, step, active, d_out, d_cyc); break;
            case RD32: hipLaunchKernelGGL(k_lds<RD32>, g, b, sm, 0, d_addr, step, active, d_out, d_cyc); break;
            case RD128: hipLaunchKernelGGL(k_lds<RD128>, g, b, sm, 0, d_addr, step, active, d_out, d_cyc); break;
            case WR64: hipLaunchKernelGGL(k_lds<WR64>, g, b, sm, 0, d_addr, step, active, d_out, d_cyc); break;
            case WR32: hipLaunchKernelGGL(k_lds<WR32>, g, b, sm, 0, d_addr, step, active, d_out, d_cyc); break;
            case WR16: hipLaunchKernelGGL(k_lds<WR16>, g, b, sm, 0, d_addr, step, active, d_out, d_cyc); break;
            case WR8: hipLaunchKernelGGL(k_lds<WR8>, g, b, sm, 0, d_addr, step, active, d_out, d_cyc); break;
            case RD8: hipLaunchKernelGGL(k_lds<RD8>, g, b, sm, 0, d_addr, step, active, d_out, d_cyc); break;
            case RDWR64: hipLaunchKernelGGL(k_lds<RDWR64>, g, b, sm, 0, d_addr, step, active, d_out, d_cyc); break;
            case RD16: hipLaunchKernelGGL(k_lds<RD16>, g, b, sm, 0, d_addr, step, active, d_out, d_cyc); break;
            }
            CK(hipDeviceSynchronize());
        }
        const int nw = ncu * wpc;
        CK(hipMemcpy(h.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> v(h.begin(), h.begin() + nw);
        const double c = median(v);
        const double per = c / ITERS;           // cycles per instruction per wave
        const double percu = per / wpc;         // CU-level cycles per wave-instruction (throughput)
        printf("LDS %-26s %-30s step %4u active %2u waves/CU %2d: %7.1f cyc/instr/wave  %6.2f cyc/instr/CU  %6.1f B/clk/CU\n", on[op], pt.name, step, active, wpc,
               per, percu, (double)active * sizes[op] * (op == RDWR64 ? 2 : 1) / percu);
    };
#define SETA(OP) CK(hipFuncSetAttribute((const void *)k_lds<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    SETA(RD64); SETA(RD32); SETA(RD128); SETA(WR64); SETA(WR32); SETA(WR16); SETA(WR8); SETA(RD8); SETA(RDWR64); SETA(RD16);
    for (int op : {RD64, RD32, RD128, RD16, RD8, WR64, WR32, WR16, WR8, RDWR64}) {
        for (const Pat &pt : pats) run_lds(op, pt, 0, 64, 16);
    }
    // partial activity and moving addresses, the interesting ops only
    for (int op : {RD64, WR64, WR32, WR8}) {
        for (uint32_t act : {32u, 16u, 8u}) run_lds(op, pats[2], 0, act, 16);
        run_lds(op, pats[2], 8, 64, 16);
        run_lds(op, pats[6], 8, 64, 16);
        run_lds(op, pats[2], 0, 64, 4);
        run_lds(op, pats[2], 0, 64, 8);
    }
    // single-wave latency of a dependent chain is not measured here (the guide gives ~64 cycles)

    // ---- far-match read-back
    {
        const uint64_t stride = 131072;
        const int waves = ncu * 16;
        uint8_t *d_buf; CK(hipMalloc(&d_buf, (size_t)waves * stride + 4096)); CK(hipMemset(d_buf, 1, (size_t)waves * stride + 4096));
        uint32_t ha[64]; for (int l = 0; l < 64; l++) ha[l] = rand();
        CK(hipMemcpy(d_addr, ha, 256, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int wpc : {4, 16}) {
            hipLaunchKernelGGL(k_gfar, dim3(ncu * 16 / wpc * wpc / wpc), dim3(64 * wpc), 0, 0, d_buf, stride, d_addr, d_out, d_cyc);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gfar, dim3(ncu), dim3(64 * wpc), 0, 0, d_buf, stride, d_addr, d_out, d_cyc);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double nreq = (double)ncu * wpc * 64 * 256;
            printf("global unaligned 8-B loads, random in a 64 KiB window per wave, %d waves/CU: %.3f ms, %.1f G lane-loads/s, %.1f cycles@2.4GHz per wave-instr per CU\n",
                   wpc, ms, nreq / ms / 1e6, ms * 1e-3 * 2.4e9 / (256.0 * wpc) );
        }
    }
    return 0;
}
