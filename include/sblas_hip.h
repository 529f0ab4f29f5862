/*
 * sblas_hip.h -- C ABI of libsblas_hip.so, the MI355X (gfx950) core of the S-BLAS CSR
 * SpMV / SpMM hot path.
 *
 * The reference (tartarughina/S-BLAS) has no FFI: its boundary for this path is the set of
 * cuSPARSE / NCCL / CUDA-runtime calls made by the header templates sblas_spmm_csr_v1/_v2 and
 * sblas_spmv_csr_v1.  Every entry point below replaces one of those call sites and takes the
 * same information the replaced call received (plain pointers and sizes, no C++ or torch
 * types).  The header-template layer in s-blas_amd/include/ (sblas.h, matrix.h, spmm.h, spmv.h)
 * forwards to these functions; INTEGRATION.md shows the binding a reference maintainer adds.
 *
 * Conventions
 *   - all array arguments of the *_hip_* compute functions are DEVICE pointers on device `dev`
 *     (dev < 0: use the calling thread's current device);
 *   - `stream` is a hipStream_t passed as void* (NULL = the device's null stream); nothing in the
 *     compute functions allocates, frees or synchronises, so they are graph-capturable;
 *   - CSR is base-0, int32 indices, fp64 values; `rowptr` is relative to the colidx/val pointers
 *     that are passed (so the re-based row-block slices of method 2 work as they are);
 *   - dense B / C are COLUMN-major with leading dimensions ldb / ldc (the only layout the
 *     reference's GPU paths accept: spmm.h:91-98, :171-178);
 *   - return value 0 = success, otherwise one of SBLAS_E_* (see sblas_hip_error_string).
 */
#ifndef SBLAS_HIP_H
#define SBLAS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBLAS_OK 0
#define SBLAS_E_INVALID 1   /* bad argument (null pointer, negative size, ld too small) */
#define SBLAS_E_HIP 2       /* a HIP runtime call or kernel launch failed              */
#define SBLAS_E_WORKSPACE 3 /* workspace missing or too small                          */
#define SBLAS_E_RCCL 4      /* RCCL unavailable or a collective failed                 */
#define SBLAS_E_IO 5        /* MatrixMarket file could not be read / parsed            */
#define SBLAS_E_NOGPU 6     /* no HIP device visible                                   */

int sblas_hip_version(void);
const char *sblas_hip_error_string(int code);
/* Number of visible HIP devices (0 when none / no driver).  Never initialises a context. */
int sblas_hip_device_count(void);

/* ---------------------------------------------------------------------------------------
 * SpMM:  C = alpha * A * B + beta * C        A: rows x cols CSR, B: cols x n, C: rows x n
 * Replaces cusparseSpMM_bufferSize + cusparseSpMM (NON_TRANSPOSE, ALG_DEFAULT, col-major) at
 *   spmm.h:134-149  (method 1: full A, B/C column block of n_i columns, real alpha/beta)
 *   spmm.h:239-251  (method 2: re-based row block A_i, C = Ccopy + start_row, ldc = M, alpha=beta=1)
 * The workspace plays the role of cuSPARSE's externalBuffer (spmm.h:140-141, :245-246): it
 * holds the row-major staging copy of B that the kernels gather from.
 * ------------------------------------------------------------------------------------- */
size_t sblas_hip_spmm_csr_f64_i32_workspace(int64_t rows, int64_t cols, int64_t nnz, int64_t n);

int sblas_hip_spmm_csr_f64_i32(int dev, void *stream,
                               int64_t rows, int64_t cols, int64_t nnz,
                               const int32_t *rowptr, const int32_t *colidx, const double *val,
                               const double *B, int64_t ldb, int64_t n,
                               double alpha, double beta,
                               double *C, int64_t ldc,
                               void *workspace, size_t workspace_bytes);

/* The two stages of the call above, exposed so that a caller that multiplies the same B more
 * than once (method 2 keeps B resident) or wants per-stage timing can drive them itself:
 *   stage 1: Bt ((cols + 1) x ldbt row-major, zero padded, last row all zero) <- B (cols x n col-major)
 *   stage 2: the row-panel SpMM kernels reading Bt.
 * ldbt = sblas_hip_spmm_ldbt(n).  Bt must be a buffer of sblas_hip_spmm_csr_f64_i32_workspace(rows, cols, nnz, n)
 * bytes: stage 2 keeps its panel verdicts behind the staging copy.  (The split form does not chunk columns:
 * (cols + 1) * ldbt * 8 must stay below 4 GiB; the one-call form walks wider B in column chunks.) */
int64_t sblas_hip_spmm_ldbt(int64_t n);
int sblas_hip_dense_to_rowmajor_f64(int dev, void *stream, int64_t cols, int64_t n,
                                    const double *B, int64_t ldb, double *Bt, int64_t ldbt);
int sblas_hip_spmm_csr_rowmajorB_f64_i32(int dev, void *stream,
                                         int64_t rows, int64_t cols, int64_t nnz,
                                         const int32_t *rowptr, const int32_t *colidx,
                                         const double *val,
                                         const double *Bt, int64_t ldbt, int64_t n,
                                         double alpha, double beta, double *C, int64_t ldc);

/* A per-matrix plan -- the slot cusparseSpMM_bufferSize / the workspace step occupy in the reference (spmm.h:134-141).
 * The unplanned call classifies A's row panels on the device on EVERY call (which stage-2 kernel computes a panel, the
 * matrix-wide votes, a row block's column range) and then launches every stage-2 kernel; those that find nothing to do
 * leave at once.  A is usually multiplied many times: sblas_hip_spmm_plan_create runs that analysis once for one
 * structure (rowptr, colidx) and one width n, keeps the verdicts in a device buffer of its own and looks at them once
 * on the host (it synchronises `stream`); a planned call then stages B and launches ONLY the kernels that have panels.
 * Results are bit-identical to the unplanned call.  The plan refers to the structure arrays it was made from: the
 * caller recreates it when their contents change (values may change freely).  One call at a time per plan (the
 * staging pass keeps its "B holds a non-finite value" flag in the plan's buffer).  A planned call allocates nothing
 * and never synchronises (graph-capturable); create / destroy do. */
int sblas_hip_spmm_plan_create(int dev, void *stream, int64_t rows, int64_t cols, int64_t nnz,
                               const int32_t *rowptr, const int32_t *colidx, int64_t n, void **plan_out);
int sblas_hip_spmm_plan_destroy(void *plan);
/* out: [0] planned at all (0: empty matrix or a pinned kernel selection -- calls run unplanned), panels for the
 * LDS-tiled [1] / direct [2] / matrix-core [3] kernels, [4] which direct kernel at 128+ staged columns (0 row per wave,
 * 1 row merging, 2 four rows per wave), [5] only the
 * block's column range of B is staged, [6] staged width, [7] rows per panel */
int sblas_hip_spmm_plan_info(const void *plan, int64_t out[8]);
int sblas_hip_spmm_csr_f64_i32_planned(const void *plan, int dev, void *stream,
                                       int64_t rows, int64_t cols, int64_t nnz,
                                       const int32_t *rowptr, const int32_t *colidx, const double *val,
                                       const double *B, int64_t ldb, int64_t n, double alpha, double beta,
                                       double *C, int64_t ldc, void *workspace, size_t workspace_bytes);

/* Diagnostics (synchronises the current device): how many row panels of the SpMM launches since the last reset
 * took the LDS-windowed path [0], the direct path because they are too sparse over their column span [1], or were
 * windowed and then recomputed by the in-kernel fallback (rows not in ascending column order) [2], or the matrix-core
 * (MFMA) path [3]. */
int sblas_hip_debug_spmm_panel_stats(uint64_t out[4], int reset);
/* Opt-in check of the CONTENTS of a CSR structure on the device (synchronises `stream`): row pointers ascending from 0 to
 * nnz, column indices inside [0, cols).  SBLAS_OK / SBLAS_E_INVALID.  The compute entry points trust the contents (as
 * cuSPARSE does); SBLAS_VALIDATE=1 makes every SpMM / SpMV call run this check first (debugging). */
int sblas_hip_debug_validate_csr_i32(int dev, void *stream, int64_t rows, int64_t cols, int64_t nnz,
                                     const int32_t *rowptr, const int32_t *colidx);
/* The library reads its experiment switches (SBLAS_SPMM_VARIANT, SBLAS_SPMV_VARIANT, ...; none changes a result) from
 * the environment once, at the first launch.  A process that changes them afterwards (the test-suite does) calls this
 * to have them read again. */
int sblas_hip_debug_reload_env(void);
/* Diagnostics for per-kernel timing (bench.py's roofline object): while enabled, the SpMM launcher brackets the
 * dominant stage-2 kernel (the LDS-windowed one) with two HIP events
 * on the launch stream; ..._last_kernel_ms waits for the second event of the most recent launch on the current device
 * and returns the elapsed time.  Off by default: a timed region is not perturbed. */
int sblas_hip_debug_spmm_kernel_events(int enable);
int sblas_hip_debug_spmm_last_kernel_ms(float *ms);

/* ---------------------------------------------------------------------------------------
 * SpMV:  y = alpha * A * x + beta * y
 * Replaces cusparseSpMV_bufferSize + cusparseSpMV at spmv.h:94-106 (no workspace is needed).
 * ------------------------------------------------------------------------------------- */
int sblas_hip_spmv_csr_f64_i32(int dev, void *stream,
                               int64_t rows, int64_t cols, int64_t nnz,
                               const int32_t *rowptr, const int32_t *colidx, const double *val,
                               const double *x, double alpha, double beta, double *y);

/* ---------------------------------------------------------------------------------------
 * y = beta * y + alpha * x   (elementwise, n elements)
 * Replaces the denseVector_plusEqual_denseVector launches at matrix.h:613-625 and :714-726
 * (kernel.h:27-38).  Note the argument order mirrors the kernel: y is updated in place.
 * ------------------------------------------------------------------------------------- */
int sblas_hip_axpby_f64(int dev, void *stream, int64_t n,
                        double alpha, const double *x, double beta, double *y);

/* ---------------------------------------------------------------------------------------
 * Partial-result merge over the GPUs of one node (RCCL over xGMI).
 * Replaces ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy at
 *   spmm.h:179-181,189,260-262,279 and spmv.h:43-45,58,115-118,134.
 * One process drives n_gpu devices (the reference's process model); the communicator set is
 * created once and reused (sblas_hip_comm_get caches per device list).  RCCL is dlopen'ed on
 * first use so that the library loads on machines without it.
 * When several ranks are mapped onto ONE physical device (devs[] all equal; used to rehearse
 * the g-way paths on a single-GPU box) the sum is done by an on-device kernel instead.
 * ------------------------------------------------------------------------------------- */
int sblas_hip_comm_get(int n_gpu, const int *devs, void **comm_out);
void sblas_hip_comm_release_all(void);
/* In-place sum all-reduce of `count` doubles; bufs[r] / streams[r] belong to rank r.
 * Issues the collective for every rank (grouped) from the calling thread. */
int sblas_hip_allreduce_sum_f64(void *comm, double *const *bufs, void *const *streams,
                                int64_t count);

/* Method-2 / SpMV merge without the all-reduce (fast path for spmm.h:222-283 and spmv.h:60-138: the zero-filled
 * M x N `C_copy`, the ncclAllReduce over all of it and the axpby launch).  The partial results of the nnz row-block
 * partition (matrix.h:356-395) are disjoint except for the rows a block boundary cuts, so rank q computes only its
 * own rows into a PACKED buffer partial[q] (num_rows[q] x N, column-major, leading dimension num_rows[q];
 * start_row[q] = starting_row_gpu[q], num_rows[q] = get_gpu_row_ptr_num(q) - 1), every rank receives the other
 * ranks' packed blocks into gather[r] (room for the sum of all blocks; own block is not copied) over RCCL
 * send/recv -- half the xGMI bytes of the all-reduce, no redundant adds -- and one kernel per rank does
 *   C_r[i, j] = beta * C_r[i, j] + alpha * sum_{q : row i in block q} partial_q[i - start_row[q], j].
 * Ranks folded onto one device skip the copies.  Stream-ordered after each rank's producers; never synchronises. */
int sblas_hip_merge_rowblocks_f64(void *comm, int64_t M, int64_t N, const int64_t *start_row,
                                  const int64_t *num_rows, double *const *partial, double *const *gather,
                                  double alpha, double beta, double *const *C, int64_t ldc, void *const *streams);
/* The scatter + alpha/beta pass alone, for callers that moved the blocks themselves (torch.distributed, MPI ...):
 * src[q] are g packed blocks resident on `device`. */
int sblas_hip_merge_rowblocks_local_f64(int device, void *stream, int64_t M, int64_t N, int g,
                                        const int64_t *start_row, const int64_t *num_rows,
                                        const double *const *src, double alpha, double beta, double *C, int64_t ldc);

/* ---------------------------------------------------------------------------------------
 * The other value / index types of the reference's templates.  sblas_spmm_csr_v1/_v2 and sblas_spmv_csr_v1 are
 * templated over <IdxType, DataType> and hand cuSPARSE getCudaDataType<float|double>() and
 * getCusparseIndexType<int32_t|int64_t>() (utility.h:302-316; spmm.h:109-118, :196-213; spmv.h:64-77, :115-118).
 * The entry points below take the two types as tags and untyped pointers; <SBLAS_I32, SBLAS_F64> forwards to the
 * tuned *_f64_i32 functions above, the other three combinations run the plain kernels of typed_kernels.hip (same
 * semantics, sums in the value type in CSR order; not tuned).  alpha / beta are passed as doubles and converted.
 * One divergence: the reference's method-2 SpMM all-reduces with ncclDouble whatever DataType is (spmm.h:260-262);
 * here the merge runs in the value type (as the reference's SpMV does, spmv.h:115-118).
 * ------------------------------------------------------------------------------------- */
#define SBLAS_F64 0
#define SBLAS_F32 1
#define SBLAS_I32 0
#define SBLAS_I64 1
size_t sblas_hip_spmm_csr_workspace(int vtype, int itype, int64_t rows, int64_t cols, int64_t nnz, int64_t n);
int sblas_hip_spmm_csr(int dev, void *stream, int vtype, int itype, int64_t rows, int64_t cols, int64_t nnz,
                       const void *rowptr, const void *colidx, const void *val, const void *B, int64_t ldb, int64_t n,
                       double alpha, double beta, void *C, int64_t ldc, void *workspace, size_t workspace_bytes);
int sblas_hip_spmv_csr(int dev, void *stream, int vtype, int itype, int64_t rows, int64_t cols, int64_t nnz,
                       const void *rowptr, const void *colidx, const void *val, const void *x, double alpha,
                       double beta, void *y);
/* kernel.h:27-38 in either value type */
int sblas_hip_axpby(int dev, void *stream, int vtype, int64_t n, double alpha, const void *x, double beta, void *y);
int sblas_hip_allreduce_sum(void *comm, int vtype, void *const *bufs, void *const *streams, int64_t count);
int sblas_hip_merge_rowblocks(void *comm, int vtype, int64_t M, int64_t N, const int64_t *start_row,
                              const int64_t *num_rows, void *const *partial, void *const *gather, double alpha,
                              double beta, void *const *C, int64_t ldc, void *const *streams);
/* sblas_partition_nnz (below) for 64-bit row pointers */
int64_t sblas_partition_nnz_i64(const int64_t *rowptr, int64_t rows, int64_t nnz, int n_gpu, int i_gpu,
                                int64_t *start_row, int64_t *stop_row, int64_t *nnz_i, int64_t *first_nnz,
                                int64_t *rebased_rowptr);

/* ---------------------------------------------------------------------------------------
 * Host-side placement arithmetic (pure functions, no GPU needed).
 * ------------------------------------------------------------------------------------- */
/* csr_findRowIdxUsingNnzIdx, utility.h:292-300 (same answer, O(log M)). */
int32_t sblas_find_row_of_nnz(const int32_t *rowptr, int32_t rows, int32_t nnz_idx);

/* nnz-balanced row-block partition of CsrSparseMatrix::sync2gpu(segment), matrix.h:356-375.
 * Uses the exact integer ceil((nnz)/g) (the reference's float ceil drops nonzeros above 2^24;
 * both agree below).  rebased_rowptr may be NULL; otherwise it must hold stop-start+2 ints.
 * Returns the number of row pointers (stop_row - start_row + 2) or a negative value. */
int64_t sblas_partition_nnz(const int32_t *rowptr, int32_t rows, int32_t nnz, int n_gpu, int i_gpu,
                            int32_t *start_row, int32_t *stop_row, int32_t *nnz_i,
                            int64_t *first_nnz, int32_t *rebased_rowptr);

/* Leading-dimension block partition of DenseMatrix::sync2gpu(segment), matrix.h:554-568. */
int sblas_partition_dense(int64_t first_order, int n_gpu, int i_gpu,
                          int64_t *offset, int64_t *dim);

/* Dense initialiser of the reference's DenseMatrix(h, w, order) / DenseVector(len) constructors (matrix.h:519-528,
 * :663-672; utility.h:197; config.h:23 seed 211): srand(seed), then rand() / RAND_MAX in storage order (host memory). */
int sblas_host_fill_rand0to1(double *dst, int64_t count, unsigned seed);

/* ---------------------------------------------------------------------------------------
 * MatrixMarket -> CSR (host).  Same observable result as mmio_info / mmio_data
 * (mmio_highlevel.h:7-127, :130-281): file order preserved inside a row, symmetric/hermitian
 * mirrored, skew treated as general, pattern -> 1.0, complex -> real part, 1-based -> 0-based.
 * Single pass over the text, arrays sized by the caller from sblas_mm_read_info.
 * ------------------------------------------------------------------------------------- */
int sblas_mm_read_info(const char *path, int32_t *rows, int32_t *cols, int32_t *nnz,
                       int32_t *is_symmetric);
int sblas_mm_read_csr(const char *path, int32_t *rowptr, int32_t *colidx, double *val);

#ifdef __cplusplus
}
#endif
#endif /* SBLAS_HIP_H */
