// micro-benchmark (round 6): how many SCATTERED store requests per second does the chip take?  Every lane writes its own
// stream (64 lanes = 64 lines per instruction), S bytes per store; and the same bytes written 4 lanes per 64-byte line.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ub_scatter profiles/scripts/ub_scatter.hip && /tmp/ub_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int S, bool COOP>
__global__ void __launch_bounds__(64) k(uint8_t *base, uint32_t per_lane, uint32_t iters)
{
    const uint32_t lane = threadIdx.x;
    uint8_t *p;
    if (COOP) p = base + ((size_t)blockIdx.x * 16 + (lane >> 2)) * per_lane * 4 + (lane & 3) * 16; // 16 streams per wave, 4 lanes a line
    else p = base + ((size_t)blockIdx.x * 64 + lane) * per_lane;
    uint4 v = make_uint4(lane, iters, 3, 4);
    for (uint32_t i = 0; i < iters; i++) {
        if (S == 4) *reinterpret_cast<uint32_t *>(p) = v.x;
        else if (S == 8) *reinterpret_cast<uint2 *>(p) = make_uint2(v.x, v.y);
        else *reinterpret_cast<uint4 *>(p) = v;
        p += COOP ? 64 : S;
        v.x += i;
    }
}
template <int S, bool COOP>
static void run(const char *name, uint8_t *d, int waves_per_cu)
{
    const uint32_t grid = 256 * waves_per_cu, iters = COOP ? 256 : 1024 * 16 / S / 4;
    const uint32_t per_lane = 16384;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<S, COOP>), dim3(grid), dim3(64), 0, 0, d, per_lane, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<S, COOP>), dim3(grid), dim3(64), 0, 0, d, per_lane, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)grid * iters, bytes = instr * 64 * (COOP ? 16 : S);
    const double req = instr * (COOP ? 16 : 64);
    printf("%-44s %2d waves/CU  %8.3f ms  %7.1f G lane-stores/s  %7.1f G line requests/s  %7.1f GB/s\n", name, waves_per_cu, ms, instr * 64 / ms / 1e6, req / ms / 1e6, bytes / ms / 1e6);
}
int main()
{
    uint8_t *d; const size_t n = (size_t)256 * 16 * 64 * 16384;
    if (hipMalloc(&d, n) != hipSuccess) return 1;
    hipMemset(d, 0, n);
    for (int w : {4, 8, 16}) {
        run<4, false>("a stream per lane, 4-byte stores", d, w);
        run<8, false>("a stream per lane, 8-byte stores", d, w);
        run<16, false>("a stream per lane, 16-byte stores", d, w);
        run<16, true>("a stream per 4 lanes, a 64-byte line per store", d, w);
    }
    return 0;
}
