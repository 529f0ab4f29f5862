// Microbenchmark (round 3): SpMM with a ROW PER LANE reading the column-major B directly -- no row-major staging copy.
// For stencil-like rows (neighbouring rows refer to neighbouring columns) the 64 lanes of a wave read 64 neighbouring
// elements of one column of B per (entry, column): coalesced, and C (column-major) is written coalesced too.
//   hipcc -O3 --offload-arch=gfx950 tools/rowlane_bench.hip -o /tmp/rowlane_bench && /tmp/rowlane_bench [rows] [per_row] [half_band] [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>

template <int JB>
__global__ __launch_bounds__(256) void rowlane_kernel(int rows, const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                                      const double *__restrict__ val, const double *__restrict__ B, long ldb,
                                                      int n, double alpha, double beta, double *__restrict__ C, long ldc)
{
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * JB;
    if (row >= rows) return;
    const int p0 = rowptr[row], p1 = rowptr[row + 1];
    double acc[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) acc[j] = 0.0;
    for (int p = p0; p < p1; ++p) {
        const int c = colidx[p];
        const double v = val[p];
        const double *b = B + c + (long)j0 * ldb;
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[j] = fma(v, b[(long)j * ldb], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < JB; ++j)
        if (j0 + j < n) {
            double *dst = C + row + (long)(j0 + j) * ldc;
            *dst = beta == 0.0 ? alpha * acc[j] : fma(beta, *dst, alpha * acc[j]);
        }
}

int main(int argc, char **argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 1000000, per = argc > 2 ? atoi(argv[2]) : 5, hb = argc > 3 ? atoi(argv[3]) : 20;
    const int n = argc > 4 ? atoi(argv[4]) : 64;
    std::vector<int> rp(rows + 1), ci;
    std::vector<double> v;
    srand(7);
    for (int r = 0; r < rows; ++r) {
        rp[r] = (int)ci.size();
        std::vector<int> cs;
        while ((int)cs.size() < per) {
            int c = r - hb + rand() % (2 * hb + 1);
            if (c < 0 || c >= rows) continue;
            if (std::find(cs.begin(), cs.end(), c) == cs.end()) cs.push_back(c);
        }
        std::sort(cs.begin(), cs.end());
        for (int c : cs) ci.push_back(c), v.push_back((rand() % 1000) / 500.0 - 1.0);
    }
    rp[rows] = (int)ci.size();
    const long nnz = ci.size();
    std::vector<double> B((size_t)rows * n), C((size_t)rows * n, 1.0);
    for (auto &x : B) x = (rand() % 1000) / 1000.0;
    int *drp, *dci;
    double *dv, *dB, *dC;
    hipMalloc(&drp, (rows + 1) * 4); hipMalloc(&dci, nnz * 4); hipMalloc(&dv, nnz * 8);
    hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8);
    hipMemcpy(drp, rp.data(), (rows + 1) * 4, hipMemcpyHostToDevice); hipMemcpy(dci, ci.data(), nnz * 4, hipMemcpyHostToDevice);
    hipMemcpy(dv, v.data(), nnz * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int jb, int steps) {
        dim3 grid((rows + 255) / 256, (n + jb - 1) / jb);
        for (int s = 0; s < steps; ++s) {
            if (jb == 8) hipLaunchKernelGGL(rowlane_kernel<8>, grid, dim3(256), 0, 0, rows, drp, dci, dv, dB, (long)rows, n, 1.0, 1.0, dC, (long)rows);
            else if (jb == 16) hipLaunchKernelGGL(rowlane_kernel<16>, grid, dim3(256), 0, 0, rows, drp, dci, dv, dB, (long)rows, n, 1.0, 1.0, dC, (long)rows);
            else hipLaunchKernelGGL(rowlane_kernel<32>, grid, dim3(256), 0, 0, rows, drp, dci, dv, dB, (long)rows, n, 1.0, 1.0, dC, (long)rows);
        }
    };
    // correctness of one step from C = 1 on a few rows
    run(16, 1);
    hipDeviceSynchronize();
    std::vector<double> got(C.size());
    hipMemcpy(got.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int r : {0, 1, rows / 2, rows - 1})
        for (int j : {0, n / 2, n - 1}) {
            double ref = 1.0;
            for (int p = rp[r]; p < rp[r + 1]; ++p) ref += v[p] * B[ci[p] + (size_t)j * rows];
            worst = std::max(worst, std::fabs(ref - got[r + (size_t)j * rows]));
        }
    const double alg = nnz * 12.0 + (rows + 1) * 4.0 + 8.0 * rows * n + 16.0 * rows * n;
    for (int jb : {8, 16, 32}) {
        run(jb, 5);
        hipEventRecord(e0, 0);
        run(jb, 20);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 20;
        printf("rows %d, %d per row, band +-%d, N = %d, %d columns per pass: %.4f ms per step = %.0f GB/s algorithmic (max |diff| of the checked entries %.2e)\n",
               rows, per, hb, n, jb, ms, alg / ms / 1e6, worst);
    }
    return 0;
}
