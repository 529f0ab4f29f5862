// Issue rate of v_mfma_f64_16x16x4_f64 against v_fma_f64 on this chip (both 78.6 TFLOP/s on paper for MI355X).
// hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_rate.hip -o tools/mfma_f64_rate && tools/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int ACC> __global__ __launch_bounds__(256) void mfma_loop(double *out, int iters, double a, double b)
{
    v4d acc[ACC];
    for (int t = 0; t < ACC; ++t) acc[t] = v4d{0, 0, 0, 0};
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int t = 0; t < ACC; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    double s = 0;
    for (int t = 0; t < ACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ACC> __global__ __launch_bounds__(256) void fma_loop(double *out, int iters, double a, double b)
{
    double acc[ACC];
    for (int t = 0; t < ACC; ++t) acc[t] = threadIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int t = 0; t < ACC; ++t) acc[t] = fma(a, acc[t], b);
    double s = 0;
    for (int t = 0; t < ACC; ++t) s += acc[t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    double *out;
    hipMalloc(&out, 256 * 16 * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int wpc = 4; wpc <= 16; wpc *= 2) { // waves per CU (blocks of 4 waves)
        const int blocks = 256 * wpc / 4;
        float ms;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1.0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        double flops = 2.0 * 16 * 16 * 4 * 4.0 * iters * blocks * 4;
        printf("mfma_f64_16x16x4  %2d waves/CU: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", wpc, ms,
               flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (4.0 * iters * wpc / 4.0));
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(fma_loop<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.000001, 1e-9);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        flops = 2.0 * 8.0 * iters * blocks * 256;
        printf("v_fma_f64         %2d waves/CU: %.3f ms  %.1f TFLOP/s\n", wpc, ms, flops / ms / 1e9);
    }
    return 0;
}
