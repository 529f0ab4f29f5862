// typed_kernels.hip -- SpMM / SpMV / merges for the value and index types the tuned path does not cover.
//
// The reference is templated over <IdxType, DataType> and maps them onto cuSPARSE's generic API with
// getCudaDataType<float|double>() and getCusparseIndexType<int32_t|int64_t>() (utility.h:302-316; used at
// spmm.h:109-118, :196-213, spmv.h:64-77).  Everything measured and tuned in this library is <int32, fp64>
// (kernels.hip, spmm_mfma.hip, spmv_kernels.hip); this file serves the other three combinations -- fp32 values
// and / or int64 indices -- with plain wave64 kernels: the same two-stage SpMM (row-major staging copy of B, then a
// wave per row with the 64 lanes along 64 columns of C), an SpMV with 16 lanes per row, and the merge / epilogue
// kernels in the value type.  Sums run in the value type in CSR order, as the reference's host loop does
// (spmm.h:59-64, spmv.h:25-30); fp32 results therefore agree with an fp32 host loop to rounding (fused
// multiply-add here), not bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "kernels.h"

namespace sblas {

namespace {

constexpr int TY_ROWS = 16; // rows of C per workgroup of the typed SpMM (four per wave, one after the other)

template <typename T> __device__ __forceinline__ T fma_t(T a, T b, T c);
template <> __device__ __forceinline__ double fma_t<double>(double a, double b, double c) { return fma(a, b, c); }
template <> __device__ __forceinline__ float fma_t<float>(float a, float b, float c) { return fmaf(a, b, c); }

// Stage 1: B (cols x n, column-major, ld = ldb) -> Bt (cols x ldbt, row-major, zero padded to ldbt).
template <typename T>
__global__ __launch_bounds__(256) void typed_stage_kernel(int64_t cols, int64_t n, const T *__restrict__ B, int64_t ldb,
                                                         T *__restrict__ Bt, int64_t ldbt)
{
    __shared__ T tile[64][33];
    const int64_t k0 = (int64_t)blockIdx.x * 32, j0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int64_t j = j0 + ty + 8 * u, k = k0 + tx;
        tile[ty + 8 * u][tx] = (j < n && k < cols) ? B[k + j * ldb] : T(0);
    }
    __syncthreads();
    const int jl = threadIdx.x & 63, kq = threadIdx.x >> 6;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int kk = kq + 4 * u;
        if (k0 + kk < cols && j0 + jl < ldbt) Bt[(k0 + kk) * ldbt + j0 + jl] = tile[jl][kk];
    }
}

// Stage 2: a wave owns a row and 64 columns of C.  The row's (column, value) pairs are read 64 at a time, one per
// lane, and handed round with shuffles; every nonzero is one coalesced 64-element read of a Bt row.
template <typename I, typename T>
__global__ __launch_bounds__(256) void typed_spmm_kernel(int64_t rows, const I *__restrict__ rowptr,
                                                        const I *__restrict__ colidx, const T *__restrict__ val,
                                                        const T *__restrict__ Bt, int64_t ldbt, int64_t n, T alpha,
                                                        T beta, T *__restrict__ C, int64_t ldc)
{
    __shared__ T ctile[64][TY_ROWS + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * TY_ROWS, col0 = (int64_t)blockIdx.y * 64;
    for (int rr = 0; rr < TY_ROWS / 4; ++rr) {
        const int r = wave * (TY_ROWS / 4) + rr;
        const int64_t row = row0 + r;
        T acc = T(0);
        if (row < rows) {
            const int64_t p0 = (int64_t)rowptr[row], p1 = (int64_t)rowptr[row + 1];
            for (int64_t p = p0; p < p1; p += 64) {
                const int cnt = (int)std::min<int64_t>(64, p1 - p);
                long long c = 0;
                T v = T(0);
                if (lane < cnt) {
                    c = (long long)colidx[p + lane];
                    v = val[p + lane];
                }
                int k = 0;
                for (; k + 4 <= cnt; k += 4) { // four Bt rows in flight, summed in CSR order
                    T b[4], vk[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const long long ck = __shfl(c, k + u, 64);
                        vk[u] = __shfl(v, k + u, 64);
                        b[u] = Bt[ck * ldbt + col0 + lane];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = fma_t<T>(vk[u], b[u], acc);
                }
                for (; k < cnt; ++k) {
                    const long long ck = __shfl(c, k, 64);
                    const T vk = __shfl(v, k, 64);
                    acc = fma_t<T>(vk, Bt[ck * ldbt + col0 + lane], acc);
                }
            }
        }
        ctile[lane][r] = acc;
    }
    __syncthreads();
    const int64_t nrows = std::min<int64_t>(TY_ROWS, rows - row0), ncols = std::min<int64_t>(64, n - col0);
    for (int idx = threadIdx.x; idx < 64 * TY_ROWS; idx += 256) {
        const int r = idx % TY_ROWS, j = idx / TY_ROWS;
        if (r < nrows && j < ncols) {
            T *dst = C + (col0 + j) * ldc + row0 + r;
            const T res = alpha * ctile[j][r];
            *dst = (beta == T(0)) ? res : fma_t<T>(beta, *dst, res); // beta = 0: C is not read
        }
    }
}

// y = alpha * A * x + beta * y, sixteen lanes per row (a DPP row), partial sums folded with shuffles.
template <typename I, typename T>
__global__ __launch_bounds__(256) void typed_spmv_kernel(int64_t rows, const I *__restrict__ rowptr,
                                                        const I *__restrict__ colidx, const T *__restrict__ val,
                                                        const T *__restrict__ x, T alpha, T beta, T *__restrict__ y)
{
    const int sub = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    T s = T(0);
    if (row < rows) {
        const int64_t p1 = (int64_t)rowptr[row + 1];
        for (int64_t p = (int64_t)rowptr[row] + sub; p < p1; p += 16) s = fma_t<T>(val[p], x[(int64_t)colidx[p]], s);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (row < rows && sub == 0) {
        const T res = alpha * s;
        y[row] = (beta == T(0)) ? res : fma_t<T>(beta, y[row], res);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void typed_scale_kernel(int64_t rows, int64_t n, T beta, T *__restrict__ C, int64_t ldc)
{
    const int64_t total = rows * n, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        T *dst = C + (i / rows) * ldc + (i % rows);
        *dst = (beta == T(0)) ? T(0) : beta * *dst;
    }
}

// y = beta*y + alpha*x (kernel.h:27-38)
template <typename T>
__global__ __launch_bounds__(256) void typed_axpby_kernel(int64_t n, T alpha, const T *__restrict__ x, T beta,
                                                         T *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = y[i] * beta + x[i] * alpha;
}

struct TypedPtrs {
    void *p[MAX_REPLICAS];
};
template <typename T> __global__ __launch_bounds__(256) void typed_sum_replicas_kernel(TypedPtrs bufs, int g, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        T s = T(0);
        for (int q = 0; q < g; ++q) s += static_cast<T *>(bufs.p[q])[i];
        for (int q = 0; q < g; ++q) static_cast<T *>(bufs.p[q])[i] = s;
    }
}

struct TypedBlocks {
    const void *src[MAX_REPLICAS];
    long long start[MAX_REPLICAS];
    long long nrows[MAX_REPLICAS];
};
// the row-block merge of kernels.hip (merge_rowblocks_kernel) in the value type
template <typename T>
__global__ __launch_bounds__(256) void typed_merge_rowblocks_kernel(long long M, long long N, int g, TypedBlocks b, T alpha,
                                                                   T beta, T *__restrict__ C, long long ldc)
{
    const long long total = M * N, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long r = i % M, j = i / M;
        T s = T(0);
        for (int q = 0; q < g; ++q) {
            const long long rel = r - b.start[q];
            if (rel >= 0 && rel < b.nrows[q]) s += static_cast<const T *>(b.src[q])[j * b.nrows[q] + rel];
        }
        T *dst = C + j * ldc + r;
        const T res = alpha * s;
        *dst = (beta == T(0)) ? res : fma_t<T>(beta, *dst, res);
    }
}

unsigned grid_for(int64_t items, int per_block)
{
    const int64_t b = (items + per_block - 1) / per_block;
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(b, 2048));
}

template <typename I, typename T>
hipError_t spmm_typed(hipStream_t s, int64_t rows, int64_t cols, int64_t nnz, const void *rowptr, const void *colidx,
                      const void *val, const void *B, int64_t ldb, int64_t n, double alpha, double beta, void *C,
                      int64_t ldc, void *ws)
{
    if (cols == 0 || nnz == 0) {
        if (beta != 1.0)
            hipLaunchKernelGGL(typed_scale_kernel<T>, dim3(grid_for(rows * n, 256)), dim3(256), 0, s, rows, n, (T)beta,
                               static_cast<T *>(C), ldc);
        return hipGetLastError();
    }
    const int64_t ldbt = typed_spmm_ldbt(n);
    T *Bt = static_cast<T *>(ws);
    hipLaunchKernelGGL(typed_stage_kernel<T>, dim3((unsigned)((cols + 31) / 32), (unsigned)(ldbt / 64)), dim3(256), 0, s, cols,
                       n, static_cast<const T *>(B), ldb, Bt, ldbt);
    hipLaunchKernelGGL((typed_spmm_kernel<I, T>), dim3((unsigned)((rows + TY_ROWS - 1) / TY_ROWS), (unsigned)(ldbt / 64)),
                       dim3(256), 0, s, rows, static_cast<const I *>(rowptr), static_cast<const I *>(colidx),
                       static_cast<const T *>(val), Bt, ldbt, n, (T)alpha, (T)beta, static_cast<T *>(C), ldc);
    return hipGetLastError();
}

template <typename I, typename T>
hipError_t spmv_typed(hipStream_t s, int64_t rows, const void *rowptr, const void *colidx, const void *val, const void *x,
                      double alpha, double beta, void *y)
{
    hipLaunchKernelGGL((typed_spmv_kernel<I, T>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, rows,
                       static_cast<const I *>(rowptr), static_cast<const I *>(colidx), static_cast<const T *>(val),
                       static_cast<const T *>(x), (T)alpha, (T)beta, static_cast<T *>(y));
    return hipGetLastError();
}

} // namespace

int64_t typed_spmm_ldbt(int64_t n) { return (n + 63) / 64 * 64; }

size_t typed_spmm_workspace(int vt, int64_t cols, int64_t n)
{
    if (cols <= 0 || n <= 0) return 0;
    return (size_t)cols * (size_t)typed_spmm_ldbt(n) * (vt == VT_F32 ? 4u : 8u);
}

hipError_t launch_typed_spmm(hipStream_t s, int vt, int it, int64_t rows, int64_t cols, int64_t nnz, const void *rowptr,
                             const void *colidx, const void *val, const void *B, int64_t ldb, int64_t n, double alpha,
                             double beta, void *C, int64_t ldc, void *ws)
{
    if (vt == VT_F32)
        return it == IT_I64 ? spmm_typed<int64_t, float>(s, rows, cols, nnz, rowptr, colidx, val, B, ldb, n, alpha, beta, C, ldc, ws)
                            : spmm_typed<int32_t, float>(s, rows, cols, nnz, rowptr, colidx, val, B, ldb, n, alpha, beta, C, ldc, ws);
    return it == IT_I64 ? spmm_typed<int64_t, double>(s, rows, cols, nnz, rowptr, colidx, val, B, ldb, n, alpha, beta, C, ldc, ws)
                        : spmm_typed<int32_t, double>(s, rows, cols, nnz, rowptr, colidx, val, B, ldb, n, alpha, beta, C, ldc, ws);
}

hipError_t launch_typed_spmv(hipStream_t s, int vt, int it, int64_t rows, const void *rowptr, const void *colidx,
                             const void *val, const void *x, double alpha, double beta, void *y)
{
    if (vt == VT_F32)
        return it == IT_I64 ? spmv_typed<int64_t, float>(s, rows, rowptr, colidx, val, x, alpha, beta, y)
                            : spmv_typed<int32_t, float>(s, rows, rowptr, colidx, val, x, alpha, beta, y);
    return it == IT_I64 ? spmv_typed<int64_t, double>(s, rows, rowptr, colidx, val, x, alpha, beta, y)
                        : spmv_typed<int32_t, double>(s, rows, rowptr, colidx, val, x, alpha, beta, y);
}

hipError_t launch_typed_axpby(hipStream_t s, int vt, int64_t n, double alpha, const void *x, double beta, void *y)
{
    if (vt == VT_F32)
        hipLaunchKernelGGL(typed_axpby_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, s, n, (float)alpha,
                           static_cast<const float *>(x), (float)beta, static_cast<float *>(y));
    else
        hipLaunchKernelGGL(typed_axpby_kernel<double>, dim3(grid_for(n, 256)), dim3(256), 0, s, n, alpha,
                           static_cast<const double *>(x), beta, static_cast<double *>(y));
    return hipGetLastError();
}

hipError_t launch_typed_sum_replicas(hipStream_t s, int vt, void *const *bufs, int g, int64_t n)
{
    TypedPtrs p{};
    for (int q = 0; q < g; ++q) p.p[q] = bufs[q];
    if (vt == VT_F32) hipLaunchKernelGGL(typed_sum_replicas_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, s, p, g, n);
    else hipLaunchKernelGGL(typed_sum_replicas_kernel<double>, dim3(grid_for(n, 256)), dim3(256), 0, s, p, g, n);
    return hipGetLastError();
}

hipError_t launch_typed_merge_rowblocks(hipStream_t s, int vt, int64_t M, int64_t N, int g, const void *const *src,
                                        const int64_t *start_row, const int64_t *num_rows, double alpha, double beta,
                                        void *C, int64_t ldc)
{
    TypedBlocks b{};
    for (int q = 0; q < g; ++q) {
        b.src[q] = src[q];
        b.start[q] = start_row[q];
        b.nrows[q] = num_rows[q];
    }
    if (vt == VT_F32)
        hipLaunchKernelGGL(typed_merge_rowblocks_kernel<float>, dim3(grid_for(M * N, 256)), dim3(256), 0, s, (long long)M,
                           (long long)N, g, b, (float)alpha, (float)beta, static_cast<float *>(C), (long long)ldc);
    else
        hipLaunchKernelGGL(typed_merge_rowblocks_kernel<double>, dim3(grid_for(M * N, 256)), dim3(256), 0, s, (long long)M,
                           (long long)N, g, b, alpha, beta, static_cast<double *>(C), (long long)ldc);
    return hipGetLastError();
}

} // namespace sblas
