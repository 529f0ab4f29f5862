// scratch microbenchmark (not product code): how fast can every CU pull 64 KiB Bt tiles from a small, L2-resident
// region into LDS?  Variants: register staging by 4 / 16 waves, LDS-DMA by 4 / 16 waves.  Prints B/clk/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int TILE_BYTES = 65536;
template <int NW, bool DMA>
__global__ __launch_bounds__(1024) void k(const char* __restrict__ src, size_t region, int tiles, double* sink)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double acc = 0;
    for (int t = 0; t < tiles; ++t) {
        const size_t base = ((size_t)(blockIdx.x * 7 + t) * TILE_BYTES) % region;
        char* buf = lds + (t & 1) * TILE_BYTES;
        if (wave < NW) {
            constexpr int PER = TILE_BYTES / 16 / (NW * 64);
            if (DMA) {
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int q = wave * 64 + lane + NW * 64 * i;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + base + (size_t)q * 16),
                        (__attribute__((address_space(3))) void*)(buf + (wave * 64 + NW * 64 * i) * 16), 16, 0, 0);
                }
            } else {
                double2 st[PER];
#pragma unroll
                for (int i = 0; i < PER; ++i) st[i] = *(const double2*)(src + base + (size_t)(wave * 64 + lane + NW * 64 * i) * 16);
#pragma unroll
                for (int i = 0; i < PER; ++i) *(double2*)(buf + (size_t)(wave * 64 + lane + NW * 64 * i) * 16) = st[i];
            }
        }
        __syncthreads();
        acc += *(double*)(buf + tid * 8);
    }
    if (acc == 12345.678) sink[0] = acc;
}
template <int NW, bool DMA> void run(const char* name, const char* d, size_t region, double* sink)
{
    hipFuncSetAttribute((const void*)k<NW, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE_BYTES);
    const int tiles = 256, blocks = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NW, DMA><<<blocks, 1024, 2 * TILE_BYTES>>>(d, region, 8, sink);
    hipEventRecord(a);
    k<NW, DMA><<<blocks, 1024, 2 * TILE_BYTES>>>(d, region, tiles, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * tiles * TILE_BYTES;
    printf("%-28s region %6.1f MB: %.3f ms  %.2f TB/s  (%.1f B/clk/CU at 2.3 GHz)\n", name, region / 1e6, ms, bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.3e9));
}
int main()
{
    char* d; double* sink; hipMalloc(&sink, 8);
    for (size_t region : {(size_t)2 << 20, (size_t)20 << 20, (size_t)40 << 20, (size_t)512 << 20}) {
        hipMalloc(&d, region); hipMemset(d, 1, region);
        run<4, false>("regs, 4 loader waves", d, region, sink);
        run<16, false>("regs, 16 loader waves", d, region, sink);
        run<4, true>("LDS-DMA, 4 loader waves", d, region, sink);
        run<16, true>("LDS-DMA, 16 loader waves", d, region, sink);
        hipFree(d);
    }
    return 0;
}
