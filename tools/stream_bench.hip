// scratch microbenchmark (not product code): what does this box stream from HBM with SpMV-like access shapes?
//   flat16 : grid-stride, 16 bytes per lane per load, two arrays (int col / double val), sum everything
//   chunk  : one wave per 4.8 KB "row" (1.6 KB of col + 3.2 KB of val), row starts read from a row_ptr array first
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void flat16(const int4 *__restrict__ c, const double2 *__restrict__ v, size_t n4, double *out)
{
    double s = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int4 a = c[i];
        const double2 x = v[2 * i], y = v[2 * i + 1];
        s += x.x * a.x + x.y * a.y + y.x * a.z + y.y * a.w;
    }
    if (s == 1.2345) out[0] = s;
}
template <int TRIPS>
__global__ __launch_bounds__(256) void chunk(const int *__restrict__ rp, const int *__restrict__ c, const double *__restrict__ v, int rows, double *out)
{
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int p0 = rp[row], p1 = rp[row + 1];
    double s0 = 0, s1 = 0;
    int p = p0 + lane;
    for (; p + 64 * (TRIPS - 1) < p1; p += 64 * TRIPS) {
#pragma unroll
        for (int u = 0; u < TRIPS; ++u) {
            if (u & 1) s1 += v[p + 64 * u] * c[p + 64 * u];
            else s0 += v[p + 64 * u] * c[p + 64 * u];
        }
    }
    for (; p < p1; p += 64) s0 += v[p] * c[p];
    if (s0 + s1 == 1.2345) out[0] = s0;
}
template <typename F> float timeit(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipEventRecord(a);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 20;
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int rows = 72000, per = 399; const size_t nnz = (size_t)rows * per;
    int *c, *rp; double *v, *out;
    hipMalloc(&c, nnz * 4 + 64); hipMalloc(&v, nnz * 8 + 64); hipMalloc(&rp, (rows + 1) * 4); hipMalloc(&out, 8);
    hipMemset(c, 0, nnz * 4); hipMemset(v, 0, nnz * 8);
    std::vector<int> h(rows + 1); for (int i = 0; i <= rows; ++i) h[i] = i * per;
    hipMemcpy(rp, h.data(), (rows + 1) * 4, hipMemcpyHostToDevice);
    const double bytes = nnz * 12.0;
    for (int g : {2048, 4096, 8192, 16384}) {
        float ms = timeit([&] { flat16<<<g, 256>>>((const int4 *)c, (const double2 *)v, nnz / 4, out); });
        printf("flat16 grid %5d: %.1f us  %.2f TB/s\n", g, ms * 1e3, bytes / ms / 1e9);
    }
    { float ms = timeit([&] { chunk<1><<<rows / 4, 256>>>(rp, c, v, rows, out); }); printf("row-per-wave, 1 slice per trip: %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9); }
    { float ms = timeit([&] { chunk<2><<<rows / 4, 256>>>(rp, c, v, rows, out); }); printf("row-per-wave, 2 slices per trip: %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9); }
    { float ms = timeit([&] { chunk<4><<<rows / 4, 256>>>(rp, c, v, rows, out); }); printf("row-per-wave, 4 slices per trip: %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9); }
    { float ms = timeit([&] { chunk<7><<<rows / 4, 256>>>(rp, c, v, rows, out); }); printf("row-per-wave, 7 slices per trip: %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9); }
    return 0;
}
