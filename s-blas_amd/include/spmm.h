// spmm.h -- C = alpha * A * B + beta * C with A in CSR and B, C dense column-major, on one or more MI355X.
//
// Same three entry points as the reference (spmm.h:29-33, :83-87, :163-167):
//   sblas_spmm_csr_cpu  single-threaded host loop (the verifier the drivers compare against)
//   sblas_spmm_csr_v1   method 1: A replicated, B and C split by columns, no communication
//   sblas_spmm_csr_v2   method 2: A split into nnz-balanced row blocks, B and C replicated,
//                       partial results merged by an RCCL all-reduce over xGMI, then C = beta*C + alpha*sum
// Host orchestration is one thread issuing asynchronous work on one stream per GPU (the reference forks
// one OpenMP thread per GPU and creates a cuSPARSE handle and an NCCL communicator inside every call); the
// arithmetic is done by the hand-written kernels behind libsblas_hip.so's C ABI.  There is no CPU fallback
// inside _v1/_v2: without a HIP device they print and exit.
#ifndef SBLAS_AMD_SPMM_H
#define SBLAS_AMD_SPMM_H

#include <assert.h>
#include <iostream>
#include <type_traits>
#include <algorithm>
#include <string.h>
#include <stdlib.h>
#include <vector>

#include "matrix.h"
#include "utility.h"

using namespace std;

// Host verifier.  Column-major C: for each row i, each column n: sum over the row's nonzeros in CSR order, then
// C = beta*C + alpha*sum (reference spmm.h:56-68).  Row-major C is computed consistently in place (the
// reference's row-major branch reads C with column-major indices, spmm.h:51-52; no GPU path accepts it anyway).
template <typename IdxType, typename DataType>
void sblas_spmm_csr_cpu(CsrSparseMatrix<IdxType, DataType> *pA, DenseMatrix<IdxType, DataType> *pB,
                        DenseMatrix<IdxType, DataType> *pC, DataType alpha, DataType beta)
{
    assert((pA->width) == (pB->height));
    assert((pA->height) == (pC->height));
    assert((pB->width) == (pC->width));
    if (pB->order == row_major) {
        cerr << "SBLAS_SPMM_CSR_CPU: B should be in column major!" << endl;
        exit(-1);
    }
    const size_t M = (size_t)pA->height, K = (size_t)pB->height, N = (size_t)pB->width;
    const bool c_col = (pC->order == col_major);
    for (size_t i = 0; i < M; ++i) {
        const IdxType lo = pA->csrRowPtr[i], hi = pA->csrRowPtr[i + 1];
        for (size_t n = 0; n < N; ++n) {
            const DataType *bcol = pB->val + n * K;
            DataType sum = 0;
            for (IdxType j = lo; j < hi; ++j) sum += pA->csrVal[j] * bcol[pA->csrColIdx[j]];
            DataType &c = c_col ? pC->val[n * M + i] : pC->val[i * N + n];
            c = beta * c + alpha * sum;
        }
    }
}

namespace sblas_detail {


template <typename IdxType, typename DataType>
inline void require_col_major(const char *who, DenseMatrix<IdxType, DataType> *pB, DenseMatrix<IdxType, DataType> *pC)
{
    if (pB->order == row_major) {
        cerr << who << ": B should be in column major!" << endl;
        exit(-1);
    }
    if (pC->order == row_major) {
        cerr << who << ": C should be in column major!" << endl;
        exit(-1);
    }
}

// One GPU's product.  <int, double> goes through a per-matrix plan (sblas_hip_spmm_plan_*: the slot cuSPARSE's
// bufferSize / workspace step has at spmm.h:134-141) from the SECOND call for this matrix, GPU and width on (a caller
// that multiplies once -- the reference's drivers -- pays nothing; making the plan synchronises the GPU's stream once);
// the plan stays in the CsrSparseMatrix until its next sync2gpu, and every later call launches only the kernels that
// have panels.  SBLAS_PLAN=0 keeps every call unplanned.
template <typename IdxType, typename DataType>
inline int spmm_on_gpu(CsrSparseMatrix<IdxType, DataType> *pA, unsigned i, void *stream, int vt, int it, int64_t m, int64_t K,
                       int64_t nnz, const DataType *B, int64_t ldb, int64_t n, double alpha, double beta, DataType *C,
                       int64_t ldc, void *ws, size_t ws_bytes)
{
    static const bool plans = [] {
        const char *e = getenv("SBLAS_PLAN");
        return !(e && e[0] == '0');
    }();
    if (plans && vt == SBLAS_F64 && it == SBLAS_I32 && pA->spmm_plan_gpu) {
        // a plan speaks for a staged width: every n with the same sblas_hip_spmm_ldbt(n) shares it (method 2's column tiles)
        const int64_t key = sblas_hip_spmm_ldbt(n);
        const bool mine = pA->spmm_plan_gpu[i] && pA->spmm_plan_n[i] == key;
        if (!mine && !pA->spmm_plan_gpu[i] && pA->spmm_plan_n[i] == -key) { // second call at this width: plan it
            const int rc = sblas_hip_spmm_plan_create(-1, stream, m, K, nnz, (const int32_t *)pA->csrRowPtr_gpu[i],
                                                      (const int32_t *)pA->csrColIdx_gpu[i], n, &pA->spmm_plan_gpu[i]);
            if (rc != SBLAS_OK) return rc;
            pA->spmm_plan_n[i] = key;
        } else if (!mine) { // first call at this width, or a width other than the plan's (a ragged last tile): unplanned
            if (!pA->spmm_plan_gpu[i]) pA->spmm_plan_n[i] = -key;
            return sblas_hip_spmm_csr(-1, stream, vt, it, m, K, nnz, pA->csrRowPtr_gpu[i], pA->csrColIdx_gpu[i], pA->csrVal_gpu[i],
                                      B, ldb, n, alpha, beta, C, ldc, ws, ws_bytes);
        }
        return sblas_hip_spmm_csr_f64_i32_planned(pA->spmm_plan_gpu[i], -1, stream, m, K, nnz, (const int32_t *)pA->csrRowPtr_gpu[i],
                                                  (const int32_t *)pA->csrColIdx_gpu[i], (const double *)pA->csrVal_gpu[i],
                                                  (const double *)B, ldb, n, alpha, beta, (double *)C, ldc, ws, ws_bytes);
    }
    return sblas_hip_spmm_csr(-1, stream, vt, it, m, K, nnz, pA->csrRowPtr_gpu[i], pA->csrColIdx_gpu[i], pA->csrVal_gpu[i], B, ldb,
                              n, alpha, beta, C, ldc, ws, ws_bytes);
}

constexpr int64_t M2_TILE = 128; // column tile of method 2's SpMM / merge pipeline

} // namespace sblas_detail

// Method 1.  Preconditions (as the reference): A.sync2gpu(g, replicate); B, C .sync2gpu(g, segment), col-major.
// On return the host copy C.val holds the complete result (each GPU's column block is copied back).
template <typename IdxType, typename DataType>
void sblas_spmm_csr_v1(CsrSparseMatrix<IdxType, DataType> *pA, DenseMatrix<IdxType, DataType> *pB,
                       DenseMatrix<IdxType, DataType> *pC, DataType alpha, DataType beta, unsigned n_gpu)
{
    assert((pA->width) == (pB->height));
    assert((pA->height) == (pC->height));
    assert((pB->width) == (pC->width));
    sblas_detail::require_col_major("SBLAS_SPMM_CSR_V1", pB, pC);
    const int vt = sblas_rt::vtype_of<DataType>("SBLAS_SPMM_CSR_V1"), it = sblas_rt::itype_of<IdxType>("SBLAS_SPMM_CSR_V1");
    assert(pA->policy == replicate && pB->policy == segment && pC->policy == segment);
    cout << "sblas_spmm_csr_v1 ready to start" << endl;
    const int64_t M = pA->height, K = pA->width, nnz = pA->nnz;
    for (unsigned i = 0; i < n_gpu; ++i) { // asynchronous: every GPU is busy before the first one finishes
        const int64_t n_i = (int64_t)pB->get_dim_gpu_num(i);
        printf("gpu-%d m:%d, n:%ld, k:%d\n", i, (int)pA->height, (long)n_i, (int)pA->width);
        if (n_i == 0) continue;
        CUDA_SAFE_CALL(cudaSetDevice((int)i));
        const size_t ws_bytes = sblas_hip_spmm_csr_workspace(vt, it, M, K, nnz, n_i);
        void *ws = sblas_rt::workspace(i, ws_bytes);
        sblas_rt::must_sblas(sblas_detail::spmm_on_gpu(pA, i, sblas_rt::stream(i), vt, it, M, K, nnz, pB->val_gpu[i], K, n_i,
                                                       (double)alpha, (double)beta, pC->val_gpu[i], M, ws, ws_bytes),
                             "sblas_hip_spmm_csr");
    }
    for (unsigned i = 0; i < n_gpu; ++i) pC->sync2cpu(i); // stream-ordered after GPU i's kernels
}

// Method 2.  Preconditions: A.sync2gpu(g, segment); B, C .sync2gpu(g, replicate), col-major.
// On return every C.val_gpu[i] holds the full result; the host copy is refreshed by C.sync2cpu(i).
template <typename IdxType, typename DataType>
void sblas_spmm_csr_v2(CsrSparseMatrix<IdxType, DataType> *pA, DenseMatrix<IdxType, DataType> *pB,
                       DenseMatrix<IdxType, DataType> *pC, DataType alpha, DataType beta, unsigned n_gpu)
{
    assert((pA->width) == (pB->height));
    assert((pA->height) == (pC->height));
    assert((pB->width) == (pC->width));
    sblas_detail::require_col_major("SBLAS_SPMM_CSR_V2", pB, pC);
    const int vt = sblas_rt::vtype_of<DataType>("SBLAS_SPMM_CSR_V2"), it = sblas_rt::itype_of<IdxType>("SBLAS_SPMM_CSR_V2");
    assert(pA->policy == segment && pB->policy == replicate && pC->policy == replicate);
    const int64_t M = pA->height, K = pA->width, N = pB->width;
    const size_t cnt = (size_t)M * (size_t)N;

    // persistent communicator over the logical GPUs (created once, not per call)
    std::vector<int> devs(n_gpu);
    for (unsigned i = 0; i < n_gpu; ++i) devs[i] = sblas_rt::physical_device(i);
    void *comm = NULL;
    sblas_rt::must_sblas(sblas_hip_comm_get((int)n_gpu, devs.data(), &comm), "sblas_hip_comm_get");

    // Merge.  Default: the row blocks are disjoint except for the rows a block boundary cuts, so every GPU computes
    // its own rows into a packed buffer, the blocks are exchanged point to point and one pass scatters them and
    // applies alpha / beta (sblas_hip_merge_rowblocks_f64: half the xGMI bytes of the all-reduce, no M*N zero fill,
    // no separate axpby).  SBLAS_MERGE=allreduce keeps the reference's pattern (spmm.h:222-283): zeroed M*N C_copy,
    // in-place sum all-reduce, axpby.
    const char *merge_mode = getenv("SBLAS_MERGE");
    const bool use_allreduce = merge_mode && !strcmp(merge_mode, "allreduce");
    std::vector<DataType *> ccopy(n_gpu, (DataType *)NULL), gather(n_gpu, (DataType *)NULL);
    std::vector<void *> streams(n_gpu), mstreams(n_gpu);
    std::vector<GPU_Timer *> timers(n_gpu);
    // two or more 128-column tiles: pipeline SpMM and merge over the tiles (SBLAS_M2_PIPELINE=0: one piece, serial)
    const char *pipe_mode = getenv("SBLAS_M2_PIPELINE");
    const bool pipelined = !use_allreduce && N >= 2 * sblas_detail::M2_TILE && !(pipe_mode && pipe_mode[0] == '0');
    std::vector<int64_t> starts(n_gpu), nrows(n_gpu);
    size_t all_blocks = 0;
    for (unsigned i = 0; i < n_gpu; ++i) {
        starts[i] = (int64_t)pA->starting_row_gpu[i];
        nrows[i] = (int64_t)pA->get_gpu_row_ptr_num(i) - 1;
        all_blocks += (size_t)nrows[i] * (size_t)N;
    }
    for (unsigned i = 0; i < n_gpu; ++i) {
        CUDA_SAFE_CALL(cudaSetDevice((int)i));
        streams[i] = sblas_rt::stream(i);
        mstreams[i] = sblas_rt::merge_stream(i);
        const int64_t m_i = nrows[i];
        const int64_t nnz_i = (int64_t)pA->nnz_gpu[i];
        const size_t ws_bytes = sblas_hip_spmm_csr_workspace(vt, it, m_i, K, nnz_i, N);
        void *ws = sblas_rt::workspace(i, ws_bytes);
        if (use_allreduce) {
            // zeroed partial-result buffer on the device (the reference uploads M*N zeros from the host);
            // A_i * B accumulated (alpha = beta = 1) at its row offset, ld = M
            ccopy[i] = (DataType *)sblas_rt::workspace(i, cnt * sizeof(DataType), sblas_rt::WS_PARTIAL);
            CUDA_SAFE_CALL(hipMemsetAsync(ccopy[i], 0, cnt * sizeof(DataType), (hipStream_t)streams[i]));
            sblas_rt::must_sblas(sblas_detail::spmm_on_gpu(pA, i, streams[i], vt, it, m_i, K, nnz_i, pB->val_gpu[i], K, N, 1.0, 1.0,
                                                           ccopy[i] + (size_t)pA->starting_row_gpu[i], M, ws, ws_bytes),
                                 "sblas_hip_spmm_csr");
        } else {
            // packed m_i x N block, beta = 0: nothing to clear
            ccopy[i] = (DataType *)sblas_rt::workspace(i, (size_t)m_i * (size_t)N * sizeof(DataType), sblas_rt::WS_PARTIAL);
            gather[i] = (DataType *)sblas_rt::workspace(i, all_blocks * sizeof(DataType), sblas_rt::WS_GATHER);
            if (!pipelined)
                sblas_rt::must_sblas(sblas_detail::spmm_on_gpu(pA, i, streams[i], vt, it, m_i, K, nnz_i, pB->val_gpu[i], K, N, 1.0, 0.0,
                                                               ccopy[i], m_i, ws, ws_bytes),
                                     "sblas_hip_spmm_csr");
        }
        timers[i] = new GPU_Timer((hipStream_t)(pipelined ? mstreams[i] : streams[i]));
        if (!pipelined) timers[i]->start_timer();
    }
    if (use_allreduce) {
        // sum of the partial C over all GPUs (RCCL over xGMI; stream-ordered after each GPU's SpMM)
        sblas_rt::must_sblas(sblas_hip_allreduce_sum(comm, vt, (void *const *)ccopy.data(), streams.data(), (int64_t)cnt),
                             "sblas_hip_allreduce_sum");
    } else if (!pipelined) {
        std::vector<void *> cptr(n_gpu);
        for (unsigned i = 0; i < n_gpu; ++i) cptr[i] = pC->val_gpu[i];
        sblas_rt::must_sblas(sblas_hip_merge_rowblocks(comm, vt, M, N, starts.data(), nrows.data(),
                                                       (void *const *)ccopy.data(), (void *const *)gather.data(),
                                                       (double)alpha, (double)beta, cptr.data(), M, streams.data()),
                             "sblas_hip_merge_rowblocks");
    } else {
        // Column-tile pipeline (the reference is fully serial, spmm.h:253-265): the SpMM of tile c + 1 runs on the compute
        // stream while tile c's blocks are exchanged and scattered on the GPU's second stream.  A tile of the packed
        // m_i x N block is contiguous (columns [c T, c T + T) at leading dimension m_i), its gather region and its
        // columns of C likewise; the terms of every element are added in the same order as in the one-piece merge.
        // (the last tile takes the remainder, 128..255 columns: every tile then runs the 128-column kernels the one-piece
        //  call runs, and the result is the same bit for bit)
        const int64_t T = sblas_detail::M2_TILE;
        const int64_t ntiles = N / T;
        size_t rows_all = 0;
        for (unsigned i = 0; i < n_gpu; ++i) rows_all += (size_t)nrows[i];
        for (int64_t c = 0; c < ntiles; ++c) {
            const int64_t c0 = c * T, Tc = (c == ntiles - 1) ? N - c0 : T;
            std::vector<void *> ptile(n_gpu), gtile(n_gpu), ctile(n_gpu);
            for (unsigned i = 0; i < n_gpu; ++i) {
                CUDA_SAFE_CALL(cudaSetDevice((int)i));
                const int64_t m_i = nrows[i], nnz_i = (int64_t)pA->nnz_gpu[i];
                const size_t ws_bytes = sblas_hip_spmm_csr_workspace(vt, it, m_i, K, nnz_i, N);
                void *ws = sblas_rt::workspace(i, ws_bytes);
                ptile[i] = ccopy[i] + (size_t)c0 * (size_t)m_i;
                gtile[i] = gather[i] + (size_t)c0 * rows_all;
                ctile[i] = pC->val_gpu[i] + (size_t)c0 * (size_t)M;
                sblas_rt::must_sblas(sblas_detail::spmm_on_gpu(pA, i, streams[i], vt, it, m_i, K, nnz_i,
                                                               pB->val_gpu[i] + (size_t)c0 * (size_t)K, K, Tc, 1.0, 0.0,
                                                               (DataType *)ptile[i], m_i, ws, ws_bytes),
                                     "sblas_hip_spmm_csr");
                hipEvent_t done = sblas_rt::event(i, (size_t)c);
                CUDA_SAFE_CALL(hipEventRecord(done, (hipStream_t)streams[i]));
                CUDA_SAFE_CALL(hipStreamWaitEvent((hipStream_t)mstreams[i], done, 0));
                if (c == 0) timers[i]->start_timer();
            }
            sblas_rt::must_sblas(sblas_hip_merge_rowblocks(comm, vt, M, Tc, starts.data(), nrows.data(), ptile.data(), gtile.data(),
                                                           (double)alpha, (double)beta, ctile.data(), M, mstreams.data()),
                                 "sblas_hip_merge_rowblocks");
        }
        for (unsigned i = 0; i < n_gpu; ++i) { // the op's result is ordered on the GPU's first stream, as without the pipeline
            CUDA_SAFE_CALL(cudaSetDevice((int)i));
            timers[i]->stop_timer();
            hipEvent_t merged = sblas_rt::event(i, (size_t)ntiles);
            CUDA_SAFE_CALL(hipEventRecord(merged, (hipStream_t)mstreams[i]));
            CUDA_SAFE_CALL(hipStreamWaitEvent((hipStream_t)streams[i], merged, 0));
        }
    }
    for (unsigned i = 0; i < n_gpu && !pipelined; ++i) {
        CUDA_SAFE_CALL(cudaSetDevice((int)i));
        timers[i]->stop_timer();
        if (use_allreduce) {
            // C = beta*C + alpha*Ccopy, same stream: no host round trip between merge and epilogue
            sblas_rt::must_sblas(sblas_hip_axpby(-1, streams[i], vt, (int64_t)cnt, (double)alpha, ccopy[i], (double)beta,
                                                 pC->val_gpu[i]),
                                 "sblas_hip_axpby");
        }
    }
    sblas_rt::sync_all(n_gpu);
    for (unsigned i = 0; i < n_gpu; ++i) {
        CUDA_SAFE_CALL(cudaSetDevice((int)i));
        cout << "GPU-" << i << " NCCL Time: " << timers[i]->measure() << " ms." << std::endl;
        delete timers[i];
    }
    CUDA_CHECK_ERROR();
}

#endif
