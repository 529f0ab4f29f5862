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
// SpMM stage-2 variants (wide form).  AUTO = the fastest measured one (round 1: panel classifier + WINDOW4 + DIRECT_DPP
// for the panels the classifier rejects); the other LDS-windowed
// row-panel x B-tile kernels (with their per-panel fallback) and the readlane kernel stay selectable for A/B runs
// and tests through SBLAS_SPMM_VARIANT.
enum { SPMM_VARIANT_AUTO = 0, SPMM_VARIANT_DIRECT = 1, SPMM_VARIANT_WINDOW_R32 = 2, SPMM_VARIANT_WINDOW_R64 = 3,
       SPMM_VARIANT_WINDOW_R128 = 4, SPMM_VARIANT_WINDOW_R64W64 = 5, SPMM_VARIANT_WINDOW_R32W128 = 6, SPMM_VARIANT_DIRECT_DPP = 7, SPMM_VARIANT_WINDOW2 = 8, SPMM_VARIANT_WINDOW3 = 9, SPMM_VARIANT_WINDOW4 = 10, SPMM_VARIANT_WINDOW5 = 11, SPMM_VARIANT_WINDOW6 = 12, SPMM_VARIANT_DIRECT_ROWS = 13 };
constexpr int SPMM_MIN_PANEL_ROWS = 32; // smallest classified panel (one int2 of workspace per panel; generation 6: 32)
hipError_t launch_spmm_rowpanel(hipStream_t s, int rows, int cols, int64_t nnz, const int *rowptr, const int *colidx,
                                const double *val, const double *Bt, int64_t ldbt, int n, double alpha,
                                double beta, double *C, int64_t ldc, int variant, int pre_epoch = 0);
hipError_t launch_stage_classify(hipStream_t s, int64_t cols, int64_t n, const double *B, int64_t ldb, double *Bt,
                                 int64_t ldbt, int rows, const int *rowptr, const int *colidx, int *epoch_out);
hipError_t panel_stats(unsigned long long out[4], bool reset);
hipError_t prof_stats(unsigned long long out[16], bool reset);
hipError_t launch_spmv(hipStream_t s, int rows, int cols, int64_t nnz, const int *rowptr, const int *colidx,
                       const double *val, const double *x, double alpha, double beta, double *y);
hipError_t launch_axpby(hipStream_t s, int64_t n, double alpha, const double *x, double beta, double *y);
hipError_t launch_sum_replicas(hipStream_t s, const ReplicaPtrs &bufs, int g, int64_t n);
void kernel_events_enable(bool on);
hipError_t kernel_events_last_ms(float *ms);
hipError_t launch_merge_rowblocks(hipStream_t s, int64_t M, int64_t N, int g, const double *const *src,
                                  const int64_t *start, const int64_t *nrows, double alpha, double beta, double *C,
                                  int64_t ldc);

} // namespace sblas
