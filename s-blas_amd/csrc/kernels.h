// kernels.h -- internal launcher interface between kernels.hip and the C-ABI layer (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sblas {

constexpr int MAX_REPLICAS = 16;
struct ReplicaPtrs {
    double *p[MAX_REPLICAS];
};

hipError_t launch_dense_to_rowmajor(hipStream_t s, int64_t cols, int64_t n, const double *B, int64_t ldb,
                                    double *Bt, int64_t ldbt);
hipError_t launch_spmm_rowpanel(hipStream_t s, int rows, const int *rowptr, const int *colidx,
                                const double *val, const double *Bt, int64_t ldbt, int n, double alpha,
                                double beta, double *C, int64_t ldc);
hipError_t launch_spmv(hipStream_t s, int rows, int64_t nnz, const int *rowptr, const int *colidx,
                       const double *val, const double *x, double alpha, double beta, double *y);
hipError_t launch_axpby(hipStream_t s, int64_t n, double alpha, const double *x, double beta, double *y);
hipError_t launch_sum_replicas(hipStream_t s, const ReplicaPtrs &bufs, int g, int64_t n);

} // namespace sblas
