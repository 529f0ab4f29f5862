// spmv.h -- y = alpha * A * x + beta * y with A in CSR, on one or more MI355X (reference spmv.h:15-19, :35-39).
//   sblas_spmv_csr_cpu  single-threaded host loop (the verifier)
//   sblas_spmv_csr_v1   A split into nnz-balanced row blocks, x and y replicated; partial y merged by an RCCL
//                       all-reduce, then y = beta*y + alpha*sum on every GPU
#ifndef SBLAS_AMD_SPMV_H
#define SBLAS_AMD_SPMV_H

#include <assert.h>
#include <iostream>
#include <stdio.h>
#include <algorithm>
#include <string.h>
#include <stdlib.h>
#include <vector>

#include "matrix.h"
#include "spmm.h"
#include "utility.h"

using namespace std;

template <typename IdxType, typename DataType>
void sblas_spmv_csr_cpu(CsrSparseMatrix<IdxType, DataType> *pA, DenseVector<IdxType, DataType> *pB,
                        DenseVector<IdxType, DataType> *pC, DataType alpha, DataType beta)
{
    assert((pA->width) == (pB->length));
    assert((pA->height) == (pC->length)); // the reference asserts x and y the same length (square only, spmv.h:21)
    for (size_t i = 0; i < (size_t)pA->height; ++i) {
        DataType sum = 0;
        for (IdxType j = pA->csrRowPtr[i]; j < pA->csrRowPtr[i + 1]; ++j)
            sum += pA->csrVal[j] * pB->val[pA->csrColIdx[j]];
        pC->val[i] = beta * pC->val[i] + alpha * sum;
    }
}

// Preconditions: A.sync2gpu(g, segment); x, y .sync2gpu(g, replicate).  On return every y.val_gpu[i] holds the
// result; refresh the host copy with y.sync2cpu(i).
template <typename IdxType, typename DataType>
void sblas_spmv_csr_v1(CsrSparseMatrix<IdxType, DataType> *pA, DenseVector<IdxType, DataType> *pB,
                       DenseVector<IdxType, DataType> *pC, DataType alpha, DataType beta, unsigned n_gpu)
{
    assert((pA->width == pB->length));
    assert((pA->height) == (pC->length));
    const int vt = sblas_rt::vtype_of<DataType>("SBLAS_SPMV_CSR_V1"), it = sblas_rt::itype_of<IdxType>("SBLAS_SPMV_CSR_V1");
    assert(pA->policy == segment && pB->policy == replicate && pC->policy == replicate);
    const int64_t M = pA->height, K = pA->width;

    std::vector<int> devs(n_gpu);
    for (unsigned i = 0; i < n_gpu; ++i) devs[i] = sblas_rt::physical_device(i);
    void *comm = NULL;
    sblas_rt::must_sblas(sblas_hip_comm_get((int)n_gpu, devs.data(), &comm), "sblas_hip_comm_get");

    // Merge as in sblas_spmm_csr_v2: packed row blocks + sblas_hip_merge_rowblocks_f64 by default,
    // SBLAS_MERGE=allreduce for the reference's zero-filled y copy + all-reduce + axpby (spmv.h:60-138).
    const char *merge_mode = getenv("SBLAS_MERGE");
    const bool use_allreduce = merge_mode && !strcmp(merge_mode, "allreduce");
    std::vector<DataType *> ycopy(n_gpu, (DataType *)NULL), gather(n_gpu, (DataType *)NULL);
    std::vector<void *> streams(n_gpu);
    std::vector<GPU_Timer *> timers(n_gpu);
    std::vector<int64_t> starts(n_gpu), nrows(n_gpu);
    size_t all_blocks = 0;
    for (unsigned i = 0; i < n_gpu; ++i) {
        starts[i] = (int64_t)pA->starting_row_gpu[i];
        nrows[i] = (int64_t)pA->get_gpu_row_ptr_num(i) - 1;
        all_blocks += (size_t)nrows[i];
    }
    for (unsigned i = 0; i < n_gpu; ++i) {
        CUDA_SAFE_CALL(cudaSetDevice((int)i));
        streams[i] = sblas_rt::stream(i);
        const int64_t m_i = nrows[i];
        if (use_allreduce) {
            ycopy[i] = (DataType *)sblas_rt::workspace(i, (size_t)M * sizeof(DataType), sblas_rt::WS_PARTIAL);
            CUDA_SAFE_CALL(hipMemsetAsync(ycopy[i], 0, (size_t)M * sizeof(DataType), (hipStream_t)streams[i]));
        } else {
            ycopy[i] = (DataType *)sblas_rt::workspace(i, (size_t)m_i * sizeof(DataType), sblas_rt::WS_PARTIAL);
            gather[i] = (DataType *)sblas_rt::workspace(i, all_blocks * sizeof(DataType), sblas_rt::WS_GATHER);
        }
        sblas_rt::must_sblas(
            sblas_hip_spmv_csr(-1, streams[i], vt, it, m_i, K, (int64_t)pA->nnz_gpu[i], pA->csrRowPtr_gpu[i],
                               pA->csrColIdx_gpu[i], pA->csrVal_gpu[i], pB->val_gpu[i], 1.0, use_allreduce ? 1.0 : 0.0,
                               use_allreduce ? ycopy[i] + (size_t)pA->starting_row_gpu[i] : ycopy[i]),
            "sblas_hip_spmv_csr");
        timers[i] = new GPU_Timer((hipStream_t)streams[i]);
        timers[i]->start_timer();
    }
    if (use_allreduce) {
        sblas_rt::must_sblas(sblas_hip_allreduce_sum(comm, vt, (void *const *)ycopy.data(), streams.data(), M),
                             "sblas_hip_allreduce_sum");
    } else {
        std::vector<void *> yptr(n_gpu);
        for (unsigned i = 0; i < n_gpu; ++i) yptr[i] = pC->val_gpu[i];
        sblas_rt::must_sblas(sblas_hip_merge_rowblocks(comm, vt, M, 1, starts.data(), nrows.data(),
                                                       (void *const *)ycopy.data(), (void *const *)gather.data(),
                                                       (double)alpha, (double)beta, yptr.data(), M, streams.data()),
                             "sblas_hip_merge_rowblocks");
    }
    for (unsigned i = 0; i < n_gpu; ++i) {
        CUDA_SAFE_CALL(cudaSetDevice((int)i));
        timers[i]->stop_timer();
        if (use_allreduce)
            sblas_rt::must_sblas(sblas_hip_axpby(-1, streams[i], vt, M, (double)alpha, ycopy[i], (double)beta, pC->val_gpu[i]),
                                 "sblas_hip_axpby");
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
