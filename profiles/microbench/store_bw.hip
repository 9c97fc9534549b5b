// What store bandwidth can a decoder reach on MI355X?  (round 5, VERDICT r04 item 1d: the `zeros` / `narrow` / `int4`
// shapes are one long overlapping match per block = a memset of the block by its wave.)
//   A  per-wave regions: wave w stores `span` contiguous bytes, 1 KiB per instruction (16 B per lane) -- the decoder's pattern
//   B  the same with non-temporal stores
//   C  classic grid-stride fill (consecutive waves store consecutive KiB)
//   D  hipMemsetAsync
//   E  per-wave regions, copy (load 16 B per lane, store 16 B per lane) for reference
// Build: hipcc --offload-arch=gfx950 -O3 -o store_bw store_bw.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NT>
__global__ void __launch_bounds__(256) k_fill_regions(uint8_t *dst, uint32_t span, uint64_t nwaves, uint32_t v, uint32_t rot)
{
    const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= nwaves) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t *p = dst + w * span + lane * 16u;
    const uint4 x = make_uint4(v, v, v, v);
    const uint32_t start = rot ? (uint32_t)((w * 2654435761ull) % (span / 1024u)) * 1024u : 0u; /* F: every wave starts at another KiB of its region */
    for (uint32_t i = 0; i < span; i += 1024u) {
        uint32_t o = start + i;
        if (o >= span) o -= span;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 y = {v, v, v, v};
        if (NT) __builtin_nontemporal_store(y, reinterpret_cast<u32x4 *>(p + o));
        else *reinterpret_cast<uint4 *>(p + o) = x;
    }
}
__global__ void __launch_bounds__(256) k_fill_stride(uint4 *dst, uint64_t n16, uint32_t v)
{
    const uint4 x = make_uint4(v, v, v, v);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = x;
}
__global__ void __launch_bounds__(256) k_copy_regions(const uint8_t *src, uint8_t *dst, uint32_t span, uint64_t nwaves)
{
    const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= nwaves) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint8_t *s = src + w * span + lane * 16u;
    uint8_t *p = dst + w * span + lane * 16u;
    for (uint32_t o = 0; o < span; o += 4096u) {
        uint4 a = *reinterpret_cast<const uint4 *>(s + o), b = *reinterpret_cast<const uint4 *>(s + o + 1024u);
        uint4 c = *reinterpret_cast<const uint4 *>(s + o + 2048u), d = *reinterpret_cast<const uint4 *>(s + o + 3072u);
        *reinterpret_cast<uint4 *>(p + o) = a; *reinterpret_cast<uint4 *>(p + o + 1024u) = b;
        *reinterpret_cast<uint4 *>(p + o + 2048u) = c; *reinterpret_cast<uint4 *>(p + o + 3072u) = d;
    }
}

template <typename F>
static double time_ms(hipStream_t s, int reps, F f)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const uint64_t nwaves = 65536;
    const uint32_t span = 131072;
    const uint64_t bytes = nwaves * span;
    uint8_t *d = nullptr, *d2 = nullptr;
    CK(hipMalloc((void **)&d, bytes)); CK(hipMalloc((void **)&d2, bytes));
    CK(hipMemset(d2, 1, bytes));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int reps = 10;
    auto gbps = [&](double ms) { return bytes / (ms * 1e-3) / 1e9; };
    for (int wpb : {1, 4}) {
        const dim3 g((uint32_t)(nwaves / wpb)), b(64 * wpb);
        double ms = time_ms(s, reps, [&] { hipLaunchKernelGGL(k_fill_regions<0>, g, b, 0, s, d, span, nwaves, 0u, 0u); });
        printf("A per-wave regions of 128 KiB, %d waves per workgroup, plain stores: %8.3f ms  %7.1f GB/s written\n", wpb, ms, gbps(ms));
        ms = time_ms(s, reps, [&] { hipLaunchKernelGGL(k_fill_regions<1>, g, b, 0, s, d, span, nwaves, 0u, 0u); });
        printf("B per-wave regions of 128 KiB, %d waves per workgroup, nt stores:    %8.3f ms  %7.1f GB/s written\n", wpb, ms, gbps(ms));
    }
    /* F: regions of 1 MiB (8 192 waves: the reference's block size), every wave from the start of its region / from a KiB of its own */
    for (uint32_t big : {1u << 20, 1u << 18}) {
        const uint64_t nw = bytes / big;
        const dim3 g((uint32_t)(nw / 4)), b(256);
        for (uint32_t rot : {0u, 1u}) {
            double ms = time_ms(s, reps, [&] { hipLaunchKernelGGL(k_fill_regions<0>, g, b, 0, s, d, big, nw, 0u, rot); });
            printf("F per-wave regions of %4u KiB, plain stores, %s: %8.3f ms  %7.1f GB/s written\n", big >> 10, rot ? "rotated start" : "from the start  ", ms, gbps(ms));
            ms = time_ms(s, reps, [&] { hipLaunchKernelGGL(k_fill_regions<1>, g, b, 0, s, d, big, nw, 0u, rot); });
            printf("F per-wave regions of %4u KiB, nt stores,    %s: %8.3f ms  %7.1f GB/s written\n", big >> 10, rot ? "rotated start" : "from the start  ", ms, gbps(ms));
        }
    }
    for (uint32_t rot : {1u}) {
        const dim3 g((uint32_t)(nwaves / 4)), b(256);
        double ms = time_ms(s, reps, [&] { hipLaunchKernelGGL(k_fill_regions<1>, g, b, 0, s, d, span, nwaves, 0u, rot); });
        printf("F per-wave regions of  128 KiB, nt stores,    rotated start: %8.3f ms  %7.1f GB/s written\n", ms, gbps(ms));
    }
    for (int blocks : {1024, 2048, 4096, 16384}) {
        double ms = time_ms(s, reps, [&] { hipLaunchKernelGGL(k_fill_stride, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint4 *>(d), bytes / 16, 0u); });
        printf("C grid-stride fill, %5d workgroups of 256:                         %8.3f ms  %7.1f GB/s written\n", blocks, ms, gbps(ms));
    }
    {
        double ms = time_ms(s, reps, [&] { CK(hipMemsetAsync(d, 0, bytes, s)); });
        printf("D hipMemsetAsync:                                                    %8.3f ms  %7.1f GB/s written\n", ms, gbps(ms));
    }
    {
        const dim3 g((uint32_t)(nwaves / 4)), b(256);
        double ms = time_ms(s, reps, [&] { hipLaunchKernelGGL(k_copy_regions, g, b, 0, s, d2, d, span, nwaves); });
        printf("E per-wave regions, copy:                                            %8.3f ms  %7.1f GB/s written (+ as much read)\n", ms, gbps(ms));
    }
    CK(hipFree(d)); CK(hipFree(d2));
    return 0;
}
