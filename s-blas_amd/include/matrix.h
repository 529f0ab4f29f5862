// matrix.h -- containers of the S-BLAS API (CSR / COO / CSC sparse, dense matrix, dense vector) with their
// multi-GPU placement, for MI355X.  Class names, public members and method names follow the reference's
// matrix.h (CSR :276-453, DenseMatrix :510-649, DenseVector :653-740, COO :118-272, CSC :457-506) so that code
// written against it compiles unchanged; the bodies are new:
//   - placement arithmetic comes from libsblas_hip.so (exact integer partitions, binary row search);
//   - uploads are hipMemcpyAsync from pinned memory on one stream per logical GPU, drained once at the end,
//     instead of blocking copies GPU after GPU;
//   - all sizes are size_t (the reference multiplies ints: matrix.h:635-636).
#ifndef SBLAS_AMD_MATRIX_H
#define SBLAS_AMD_MATRIX_H

#include <assert.h>
#include <iostream>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

#include "kernel.h"
#include "mmio.h"
#include "mmio_highlevel.h"
#include "utility.h"

using namespace std; // the reference injects it (matrix.h:28) and its drivers rely on it

// how a container is placed on the GPUs: host only / a full copy per GPU / one block per GPU
enum GpuSharePolicy { none = 0, replicate = 1, segment = 2 };
enum MajorOrder { row_major = 0, col_major = 1 };

namespace sblas_detail {

template <typename T> T *to_device_async(unsigned gpu, const T *host, size_t count)
{
    T *d = NULL;
    CUDA_SAFE_CALL(cudaSetDevice((int)gpu));
    SAFE_ALOC_GPU(d, count * sizeof(T));
    if (count)
        CUDA_SAFE_CALL(hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, sblas_rt::stream(gpu)));
    return d;
}

// `replicate` placement of one host array on g GPUs (reference matrix.h:331-355, :546-553: g blocking H2D copies, one
// GPU after the other).  Default: g asynchronous H2D copies, one per GPU stream, all in flight at once -- every GPU of
// an MI355X node has its own PCIe 5 x16 link.  SBLAS_REPLICATE=p2p: ONE H2D copy to GPU 0, then GPU 0 -> GPU i over
// xGMI (hipMemcpyPeerAsync on stream i behind an event on stream 0; the mesh gives every pair its own link), for
// hosts where the PCIe side (one switch, one memory channel) is the limit.  Unmeasured: the pool has one-GPU boxes
// only, where it degenerates to device-to-device copies.
template <typename T> void replicate_async(unsigned g, const T *host, size_t count, T **out)
{
    static const bool p2p = [] {
        const char *e = getenv("SBLAS_REPLICATE");
        return e && !strcmp(e, "p2p");
    }();
    if (!p2p || g < 2 || count == 0) {
        for (unsigned i = 0; i < g; ++i) out[i] = to_device_async(i, host, count);
        return;
    }
    out[0] = to_device_async(0, host, count);
    hipEvent_t landed;
    CUDA_SAFE_CALL(cudaSetDevice(0));
    sblas_rt::must(hipEventCreateWithFlags(&landed, hipEventDisableTiming), "hipEventCreate");
    sblas_rt::must(hipEventRecord(landed, sblas_rt::stream(0)), "hipEventRecord");
    for (unsigned i = 1; i < g; ++i) {
        CUDA_SAFE_CALL(cudaSetDevice((int)i));
        SAFE_ALOC_GPU(out[i], count * sizeof(T));
        sblas_rt::must(hipStreamWaitEvent(sblas_rt::stream(i), landed, 0), "hipStreamWaitEvent");
        sblas_rt::must(hipMemcpyPeerAsync(out[i], sblas_rt::physical_device(i), out[0], sblas_rt::physical_device(0),
                                          count * sizeof(T), sblas_rt::stream(i)),
                       "hipMemcpyPeerAsync");
    }
    sblas_rt::must(hipEventDestroy(landed), "hipEventDestroy"); // (released once the recorded work has completed)
}

// exact block partition of `total` leading-dimension units over g GPUs (DenseMatrix segment policy)
inline void dense_block(size_t total, unsigned g, unsigned i, size_t &offset, size_t &dim)
{
    int64_t o = 0, d = 0;
    sblas_rt::must_sblas(sblas_partition_dense((int64_t)total, (int)g, (int)i, &o, &d), "sblas_partition_dense");
    offset = (size_t)o;
    dim = (size_t)d;
}

} // namespace sblas_detail

// ----------------------------------------------------------------------------------------------
// format conversions on the host (reference matrix.h:39-91)
// ----------------------------------------------------------------------------------------------
template <typename IdxType, typename DataType>
void CsrToCsc(const IdxType m, const IdxType n, const IdxType nnz, const IdxType *csrRowPtr,
              const IdxType *csrColIdx, const DataType *csrVal, IdxType *cscRowIdx, IdxType *cscColPtr,
              DataType *cscVal)
{
    std::fill(cscColPtr, cscColPtr + (size_t)n + 1, (IdxType)0);
    for (size_t k = 0; k < (size_t)nnz; ++k) cscColPtr[csrColIdx[k]]++;
    exclusive_scan<size_t, IdxType>(cscColPtr, (size_t)n + 1);
    std::vector<IdxType> next(cscColPtr, cscColPtr + (size_t)n);
    for (IdxType r = 0; r < m; ++r)
        for (IdxType k = csrRowPtr[r]; k < csrRowPtr[r + 1]; ++k) {
            const IdxType at = next[csrColIdx[k]]++;
            cscRowIdx[at] = r;
            cscVal[at] = csrVal[k];
        }
}

template <typename IdxType, typename DataType>
void CscToCsr(const IdxType n, const IdxType m, const IdxType nnz, const IdxType *cscColPtr,
              const IdxType *cscRowIdx, const DataType *cscVal, IdxType *csrColIdx, IdxType *csrRowPtr,
              DataType *csrVal)
{
    std::fill(csrRowPtr, csrRowPtr + (size_t)m + 1, (IdxType)0);
    for (size_t k = 0; k < (size_t)nnz; ++k) csrRowPtr[cscRowIdx[k]]++;
    exclusive_scan<size_t, IdxType>(csrRowPtr, (size_t)m + 1);
    std::vector<IdxType> next(csrRowPtr, csrRowPtr + (size_t)m);
    for (IdxType c = 0; c < n; ++c)
        for (IdxType k = cscColPtr[c]; k < cscColPtr[c + 1]; ++k) {
            const IdxType at = next[cscRowIdx[k]]++;
            csrColIdx[at] = c;
            csrVal[at] = cscVal[k];
        }
}

// ----------------------------------------------------------------------------------------------
// COO (never consumed by an op; kept so that container code keeps compiling)
// ----------------------------------------------------------------------------------------------
template <typename IdxType, typename DataType> struct CooElement {
    IdxType row;
    IdxType col;
    DataType val;
};

template <typename IdxType, typename DataType> class CooSparseMatrix {
  public:
    CooSparseMatrix()
        : cooRowIdx(NULL), cooColIdx(NULL), cooVal(NULL), cooRowIdx_gpu(NULL), cooColIdx_gpu(NULL),
          cooVal_gpu(NULL), nnz_gpu(NULL), nnz(0), height(0), width(0), n_gpu(0), policy(none)
    {
    }
    // Entries exactly as listed in the file (no symmetric expansion), then sorted by (row, col).
    // Unlike the reference (matrix.h:143-175, "%d %d %lg" for every field type) pattern / integer / complex
    // files are parsed according to their banner.
    CooSparseMatrix(const char *filename, unsigned _n_gpu = 0, enum GpuSharePolicy _policy = none)
        : cooRowIdx(NULL), cooColIdx(NULL), cooVal(NULL), cooRowIdx_gpu(NULL), cooColIdx_gpu(NULL),
          cooVal_gpu(NULL), nnz_gpu(NULL), nnz(0), height(0), width(0), n_gpu(_n_gpu), policy(_policy)
    {
        cout << "Loading input matrix from '" << filename << "'." << endl;
        FILE *f = fopen(filename, "r");
        if (f == NULL) {
            cerr << "Error openning file " << filename << endl;
            exit(-1);
        }
        MM_typecode code;
        if (mm_read_banner(f, &code) != 0) {
            cerr << "Could not process Matrix Market banner." << endl;
            exit(-1);
        }
        int m, n, nz;
        if (mm_read_mtx_crd_size(f, &m, &n, &nz) != 0) {
            cerr << "Error reading matrix crd size." << endl;
            exit(-1);
        }
        height = (IdxType)m;
        width = (IdxType)n;
        nnz = (IdxType)nz;
        cout << "Height: " << height << " Width: " << width << " nnz: " << nnz << endl;
        SAFE_ALOC_HOST(cooRowIdx, get_nnz_idx_size());
        SAFE_ALOC_HOST(cooColIdx, get_nnz_idx_size());
        SAFE_ALOC_HOST(cooVal, get_nnz_val_size());
        for (size_t k = 0; k < (size_t)nz; ++k) {
            int r = 1, c = 1, iv = 0;
            double re = 1.0, im = 0.0;
            int ok;
            if (mm_is_real(code)) ok = fscanf(f, "%d %d %lg", &r, &c, &re) == 3;
            else if (mm_is_complex(code)) ok = fscanf(f, "%d %d %lg %lg", &r, &c, &re, &im) == 4;
            else if (mm_is_integer(code)) { ok = fscanf(f, "%d %d %d", &r, &c, &iv) == 3; re = iv; }
            else ok = fscanf(f, "%d %d", &r, &c) == 2;
            if (!ok) {
                cerr << "Error reading matrix entries." << endl;
                exit(-1);
            }
            cooRowIdx[k] = (IdxType)(r - 1);
            cooColIdx[k] = (IdxType)(c - 1);
            cooVal[k] = (DataType)re;
        }
        fclose(f);
        sortByRow();
        if (n_gpu != 0 && policy != none) place_on_gpus();
    }
    ~CooSparseMatrix()
    {
        SAFE_FREE_HOST(cooRowIdx);
        SAFE_FREE_HOST(cooColIdx);
        SAFE_FREE_HOST(cooVal);
        if (n_gpu != 0 && policy != none) {
            SAFE_FREE_MULTI_GPU(cooRowIdx_gpu, n_gpu);
            SAFE_FREE_MULTI_GPU(cooColIdx_gpu, n_gpu);
            SAFE_FREE_MULTI_GPU(cooVal_gpu, n_gpu);
            SAFE_FREE_HOST(nnz_gpu);
        }
    }
    void sortByRow()
    {
        std::vector<CooElement<IdxType, DataType>> e((size_t)nnz);
        for (size_t k = 0; k < (size_t)nnz; ++k) e[k] = {cooRowIdx[k], cooColIdx[k], cooVal[k]};
        std::stable_sort(e.begin(), e.end(), [](const CooElement<IdxType, DataType> &a,
                                                const CooElement<IdxType, DataType> &b) {
            return a.row != b.row ? a.row < b.row : a.col < b.col;
        });
        for (size_t k = 0; k < (size_t)nnz; ++k) {
            cooRowIdx[k] = e[k].row;
            cooColIdx[k] = e[k].col;
            cooVal[k] = e[k].val;
        }
    }
    size_t get_gpu_nnz_idx_size(unsigned i_gpu)
    {
        assert(i_gpu < n_gpu);
        return nnz_gpu ? (size_t)nnz_gpu[i_gpu] * sizeof(IdxType) : 0;
    }
    size_t get_gpu_nnz_val_size(unsigned i_gpu)
    {
        assert(i_gpu < n_gpu);
        return nnz_gpu ? (size_t)nnz_gpu[i_gpu] * sizeof(DataType) : 0;
    }
    size_t get_nnz_idx_size() { return (size_t)nnz * sizeof(IdxType); }
    size_t get_nnz_val_size() { return (size_t)nnz * sizeof(DataType); }

  private:
    void place_on_gpus()
    {
        SAFE_ALOC_HOST(cooRowIdx_gpu, n_gpu * sizeof(IdxType *));
        SAFE_ALOC_HOST(cooColIdx_gpu, n_gpu * sizeof(IdxType *));
        SAFE_ALOC_HOST(cooVal_gpu, n_gpu * sizeof(DataType *));
        SAFE_ALOC_HOST(nnz_gpu, n_gpu * sizeof(IdxType));
        for (unsigned i = 0; i < n_gpu; ++i) {
            size_t off = 0, cnt = (size_t)nnz;
            if (policy == segment) sblas_detail::dense_block((size_t)nnz, n_gpu, i, off, cnt);
            nnz_gpu[i] = (IdxType)cnt;
            cooRowIdx_gpu[i] = sblas_detail::to_device_async(i, cooRowIdx + off, cnt);
            cooColIdx_gpu[i] = sblas_detail::to_device_async(i, cooColIdx + off, cnt);
            cooVal_gpu[i] = sblas_detail::to_device_async(i, cooVal + off, cnt);
        }
        sblas_rt::sync_all(n_gpu);
    }

  public:
    IdxType *cooRowIdx;
    IdxType *cooColIdx;
    DataType *cooVal;
    IdxType **cooRowIdx_gpu;
    IdxType **cooColIdx_gpu;
    DataType **cooVal_gpu;
    IdxType *nnz_gpu;
    IdxType nnz;
    IdxType height;
    IdxType width;
    unsigned n_gpu;
    enum GpuSharePolicy policy;
};

// ----------------------------------------------------------------------------------------------
// CSR -- the operand of every op on the hot path
// ----------------------------------------------------------------------------------------------
template <typename IdxType, typename DataType> class CsrSparseMatrix {
  public:
    CsrSparseMatrix()
        : csrRowPtr(NULL), csrColIdx(NULL), csrVal(NULL), csrRowPtr_gpu(NULL), csrColIdx_gpu(NULL),
          csrVal_gpu(NULL), nnz_gpu(NULL), starting_row_gpu(NULL), stoping_row_gpu(NULL), nnz(0), height(0),
          width(0), n_gpu(0), policy(none), spmm_plan_gpu(NULL), spmm_plan_n(NULL)
    {
    }
    // MatrixMarket file -> host CSR (one parse of the text; rows keep file order, see sblas_mm_read_csr)
    CsrSparseMatrix(const char *filename)
        : csrRowPtr(NULL), csrColIdx(NULL), csrVal(NULL), csrRowPtr_gpu(NULL), csrColIdx_gpu(NULL),
          csrVal_gpu(NULL), nnz_gpu(NULL), starting_row_gpu(NULL), stoping_row_gpu(NULL), nnz(0), height(0),
          width(0), n_gpu(0), policy(none), spmm_plan_gpu(NULL), spmm_plan_n(NULL)
    {
        int m = 0, n = 0, nnzA = 0, sym = 0;
        const int rc = mmio_info(&m, &n, &nnzA, &sym, filename);
        height = (IdxType)m;
        width = (IdxType)n;
        nnz = (IdxType)nnzA;
        SAFE_ALOC_HOST(csrRowPtr, get_row_ptr_size());
        SAFE_ALOC_HOST(csrColIdx, get_col_idx_size());
        SAFE_ALOC_HOST(csrVal, get_val_size());
        csrRowPtr[0] = 0;
        if (rc != 0) return; // as the reference: a missing file leaves an empty matrix behind (matrix.h:302)
        if (std::is_same<IdxType, int>::value && std::is_same<DataType, double>::value) {
            mmio_data((int *)csrRowPtr, (int *)csrColIdx, (double *)csrVal, filename);
        } else {
            std::vector<int> rp((size_t)m + 1), ci((size_t)nnzA + 1);
            std::vector<double> v((size_t)nnzA + 1);
            mmio_data(rp.data(), ci.data(), v.data(), filename);
            for (size_t i = 0; i <= (size_t)m; ++i) csrRowPtr[i] = (IdxType)rp[i];
            for (size_t k = 0; k < (size_t)nnzA; ++k) {
                csrColIdx[k] = (IdxType)ci[k];
                csrVal[k] = (DataType)v[k];
            }
        }
        printf("input matrix A: ( %i, %i ) nnz = %i\n", m, n, nnzA);
    }
    // the per-GPU SpMM plans (made by sblas_spmm_csr_v1 / _v2 on first use) describe the device copies: gone with them
    void drop_spmm_plans()
    {
        if (spmm_plan_gpu) {
            for (unsigned i = 0; i < n_gpu; ++i)
                if (spmm_plan_gpu[i]) (void)sblas_hip_spmm_plan_destroy(spmm_plan_gpu[i]);
            free(spmm_plan_gpu);
            free(spmm_plan_n);
            spmm_plan_gpu = NULL;
            spmm_plan_n = NULL;
        }
    }
    ~CsrSparseMatrix()
    {
        drop_spmm_plans();
        SAFE_FREE_HOST(csrRowPtr);
        SAFE_FREE_HOST(csrColIdx);
        SAFE_FREE_HOST(csrVal);
        SAFE_FREE_MULTI_GPU(csrRowPtr_gpu, n_gpu);
        SAFE_FREE_MULTI_GPU(csrColIdx_gpu, n_gpu);
        SAFE_FREE_MULTI_GPU(csrVal_gpu, n_gpu);
        SAFE_FREE_HOST(nnz_gpu);
        SAFE_FREE_HOST(starting_row_gpu);
        SAFE_FREE_HOST(stoping_row_gpu);
    }
    // replicate: every GPU gets the whole matrix (method 1).
    // segment  : GPU i gets nonzeros [i*avg, min((i+1)*avg, nnz)), avg = ceil(nnz/g), with row pointers re-based
    //            to its slice; a row cut by a boundary is shared by two GPUs (method 2, SpMV).
    void sync2gpu(unsigned _n_gpu, enum GpuSharePolicy _policy)
    {
        drop_spmm_plans(); // (of the previous placement)
        n_gpu = _n_gpu;
        policy = _policy;
        assert(n_gpu != 0);
        assert(policy != none);
        spmm_plan_gpu = (void **)calloc(n_gpu, sizeof(void *));
        spmm_plan_n = (int64_t *)calloc(n_gpu, sizeof(int64_t));
        SAFE_ALOC_HOST(csrRowPtr_gpu, n_gpu * sizeof(IdxType *));
        SAFE_ALOC_HOST(csrColIdx_gpu, n_gpu * sizeof(IdxType *));
        SAFE_ALOC_HOST(csrVal_gpu, n_gpu * sizeof(DataType *));
        for (unsigned i = 0; i < n_gpu; ++i) csrRowPtr_gpu[i] = NULL, csrColIdx_gpu[i] = NULL, csrVal_gpu[i] = NULL;
        if (policy == replicate) {
            sblas_detail::replicate_async(n_gpu, csrRowPtr, (size_t)height + 1, csrRowPtr_gpu);
            sblas_detail::replicate_async(n_gpu, csrColIdx, (size_t)nnz, csrColIdx_gpu);
            sblas_detail::replicate_async(n_gpu, csrVal, (size_t)nnz, csrVal_gpu);
        } else if (policy == segment) {
            SAFE_ALOC_HOST(nnz_gpu, n_gpu * sizeof(IdxType));
            SAFE_ALOC_HOST(starting_row_gpu, n_gpu * sizeof(IdxType));
            SAFE_ALOC_HOST(stoping_row_gpu, n_gpu * sizeof(IdxType));
            std::vector<IdxType *> rebased(n_gpu, (IdxType *)NULL); // pinned staging, freed after the drain
            // the partition arithmetic lives behind the C ABI, once per index width
            auto split = [&](unsigned i, int64_t *s, int64_t *e, int64_t *k, int64_t *first, IdxType *out) -> int64_t {
                if (sizeof(IdxType) == 4) {
                    int32_t s4 = 0, e4 = 0, k4 = 0;
                    const int64_t num = sblas_partition_nnz((const int32_t *)csrRowPtr, (int32_t)height, (int32_t)nnz,
                                                            (int)n_gpu, (int)i, &s4, &e4, &k4, first, (int32_t *)out);
                    *s = s4, *e = e4, *k = k4;
                    return num;
                }
                return sblas_partition_nnz_i64((const int64_t *)csrRowPtr, (int64_t)height, (int64_t)nnz, (int)n_gpu, (int)i,
                                               s, e, k, first, (int64_t *)out);
            };
            for (unsigned i = 0; i < n_gpu; ++i) {
                int64_t s = 0, e = 0, k = 0;
                int64_t first = 0;
                const int64_t num = split(i, &s, &e, &k, &first, (IdxType *)NULL);
                if (num < 0) {
                    fprintf(stderr, "S-BLAS: cannot split %ld nonzeros over %u GPUs (GPU %u would own none)\n",
                            (long)nnz, n_gpu, i);
                    exit(-1);
                }
                starting_row_gpu[i] = (IdxType)s;
                stoping_row_gpu[i] = (IdxType)e;
                nnz_gpu[i] = (IdxType)k;
                SAFE_ALOC_HOST(rebased[i], (size_t)num * sizeof(IdxType));
                split(i, &s, &e, &k, &first, rebased[i]);
                csrRowPtr_gpu[i] = sblas_detail::to_device_async(i, rebased[i], (size_t)num);
                csrColIdx_gpu[i] = sblas_detail::to_device_async(i, csrColIdx + first, (size_t)k);
                csrVal_gpu[i] = sblas_detail::to_device_async(i, csrVal + first, (size_t)k);
                printf("gpu-%d,start-row:%d,stop-row:%d,num-rows:%ld,num-nnz:%d\n", i, (int)starting_row_gpu[i],
                       (int)stoping_row_gpu[i], (long)get_gpu_row_ptr_num(i), (int)nnz_gpu[i]);
            }
            sblas_rt::sync_all(n_gpu);
            for (unsigned i = 0; i < n_gpu; ++i) SAFE_FREE_HOST(rebased[i]);
            return;
        }
        sblas_rt::sync_all(n_gpu);
    }
    size_t get_gpu_row_ptr_num(unsigned i_gpu)
    {
        assert(i_gpu < n_gpu);
        return nnz_gpu ? (size_t)(stoping_row_gpu[i_gpu] - starting_row_gpu[i_gpu] + 2) : 0;
    }
    size_t get_gpu_row_ptr_size(unsigned i_gpu) { return get_gpu_row_ptr_num(i_gpu) * sizeof(IdxType); }
    size_t get_gpu_col_idx_num(unsigned i_gpu)
    {
        assert(i_gpu < n_gpu);
        return nnz_gpu ? (size_t)nnz_gpu[i_gpu] : 0;
    }
    size_t get_gpu_col_idx_size(unsigned i_gpu) { return get_gpu_col_idx_num(i_gpu) * sizeof(IdxType); }
    size_t get_gpu_nnz_val_num(unsigned i_gpu) { return get_gpu_col_idx_num(i_gpu); }
    size_t get_gpu_nnz_val_size(unsigned i_gpu) { return get_gpu_nnz_val_num(i_gpu) * sizeof(DataType); }
    size_t get_row_ptr_size() { return ((size_t)height + 1) * sizeof(IdxType); }
    size_t get_col_idx_size() { return (size_t)nnz * sizeof(IdxType); }
    size_t get_val_size() { return (size_t)nnz * sizeof(DataType); }

  public:
    IdxType *csrRowPtr;
    IdxType *csrColIdx;
    DataType *csrVal;
    IdxType **csrRowPtr_gpu;
    IdxType **csrColIdx_gpu;
    DataType **csrVal_gpu;
    IdxType *nnz_gpu;          // nonzeros owned by each GPU (segment policy)
    IdxType *starting_row_gpu; // first / last row touched by each GPU's slice
    IdxType *stoping_row_gpu;
    IdxType nnz;
    IdxType height;
    IdxType width;
    unsigned n_gpu;
    enum GpuSharePolicy policy;
    // (not in the reference) per-GPU plan of the SpMM ops and the width it was made for; see spmm.h
    void **spmm_plan_gpu;
    int64_t *spmm_plan_n;
};

// ----------------------------------------------------------------------------------------------
// CSC (host only; built from a CSR)
// ----------------------------------------------------------------------------------------------
template <typename IdxType, typename DataType> class CscSparseMatrix {
  public:
    CscSparseMatrix() : cscRowIdx(NULL), cscColPtr(NULL), cscVal(NULL), nnz(0), height(0), width(0) {}
    CscSparseMatrix(const CsrSparseMatrix<IdxType, DataType> *csr)
        : cscRowIdx(NULL), cscColPtr(NULL), cscVal(NULL), nnz(csr->nnz), height(csr->height), width(csr->width)
    {
        cout << "Building csc matrix from a csr matrix." << endl;
        cout << "Height: " << height << " Width: " << width << " nnz: " << nnz << endl;
        SAFE_ALOC_HOST(cscColPtr, get_col_ptr_size());
        SAFE_ALOC_HOST(cscRowIdx, get_row_idx_size());
        SAFE_ALOC_HOST(cscVal, get_val_size());
        CsrToCsc<IdxType, DataType>(height, width, nnz, csr->csrRowPtr, csr->csrColIdx, csr->csrVal, cscRowIdx,
                                    cscColPtr, cscVal);
    }
    ~CscSparseMatrix()
    {
        SAFE_FREE_HOST(cscColPtr);
        SAFE_FREE_HOST(cscRowIdx);
        SAFE_FREE_HOST(cscVal);
    }
    size_t get_col_ptr_size() { return ((size_t)width + 1) * sizeof(IdxType); }
    size_t get_row_idx_size() { return (size_t)nnz * sizeof(IdxType); }
    size_t get_val_size() { return (size_t)nnz * sizeof(DataType); }

  public:
    IdxType *cscRowIdx;
    IdxType *cscColPtr;
    DataType *cscVal;
    IdxType nnz;
    IdxType height;
    IdxType width;
};

// ----------------------------------------------------------------------------------------------
// Dense matrix
// ----------------------------------------------------------------------------------------------
template <typename IdxType, typename DataType> class DenseMatrix {
  public:
    DenseMatrix() : height(0), width(0), val(NULL), val_gpu(NULL), dim_gpu(NULL), n_gpu(0), policy(none), order(row_major) {}
    // uniform [0,1] fill: srand(RAND_INIT_SEED), rand()/RAND_MAX in storage order (bit-identical to the
    // reference's DenseMatrix(h, w, order), matrix.h:519-528, on the same libc)
    DenseMatrix(IdxType _height, IdxType _width, enum MajorOrder _order)
        : height(_height), width(_width), val(NULL), val_gpu(NULL), dim_gpu(NULL), n_gpu(0), policy(none), order(_order)
    {
        SAFE_ALOC_HOST(val, get_mtx_size());
        srand(RAND_INIT_SEED);
        const size_t cnt = get_mtx_num();
        for (size_t i = 0; i < cnt; ++i) val[i] = (DataType)rand0to1();
    }
    DenseMatrix(IdxType _height, IdxType _width, DataType _val, enum MajorOrder _order)
        : height(_height), width(_width), val(NULL), val_gpu(NULL), dim_gpu(NULL), n_gpu(0), policy(none), order(_order)
    {
        SAFE_ALOC_HOST(val, get_mtx_size());
        srand(RAND_INIT_SEED); // the reference reseeds here too (matrix.h:533)
        const size_t cnt = get_mtx_num();
        for (size_t i = 0; i < cnt; ++i) val[i] = _val;
    }
    ~DenseMatrix()
    {
        SAFE_FREE_HOST(val);
        SAFE_FREE_HOST(dim_gpu);
        SAFE_FREE_MULTI_GPU(val_gpu, n_gpu);
    }
    // replicate: full copy per GPU.  segment: blocks of ceil(first/g) leading-dimension units (columns of a
    // column-major matrix), contiguous in storage.
    void sync2gpu(unsigned _n_gpu, enum GpuSharePolicy _policy)
    {
        n_gpu = _n_gpu;
        policy = _policy;
        assert(n_gpu != 0);
        assert(policy != none);
        SAFE_ALOC_HOST(val_gpu, n_gpu * sizeof(DataType *));
        if (policy == replicate) {
            sblas_detail::replicate_async(n_gpu, val, get_mtx_num(), val_gpu);
        } else {
            SAFE_ALOC_HOST(dim_gpu, n_gpu * sizeof(IdxType));
            const size_t first = (order == row_major) ? (size_t)height : (size_t)width;
            const size_t second = (order == row_major) ? (size_t)width : (size_t)height;
            for (unsigned i = 0; i < n_gpu; ++i) {
                size_t off = 0, dim = 0;
                sblas_detail::dense_block(first, n_gpu, i, off, dim);
                dim_gpu[i] = (IdxType)dim;
                val_gpu[i] = sblas_detail::to_device_async(i, val + off * second, dim * second);
            }
        }
        sblas_rt::sync_all(n_gpu);
    }
    // host transpose into a new object of the opposite order (only without GPU copies)
    DenseMatrix *transpose()
    {
        assert(n_gpu == 0);
        assert(policy == none);
        DenseMatrix *t = new DenseMatrix(height, width, (DataType)0, (order == row_major ? col_major : row_major));
        const size_t h = (size_t)height, w = (size_t)width;
        if (order == row_major) {
            for (size_t i = 0; i < h; ++i)
                for (size_t j = 0; j < w; ++j) t->val[j * h + i] = val[i * w + j];
        } else {
            for (size_t i = 0; i < h; ++i)
                for (size_t j = 0; j < w; ++j) t->val[i * w + j] = val[j * h + i];
        }
        return t;
    }
    // device -> host of GPU i's block (segment) or of its full copy (replicate); blocking
    void sync2cpu(unsigned i_gpu)
    {
        assert(val_gpu != NULL);
        assert(i_gpu < n_gpu);
        CUDA_SAFE_CALL(cudaSetDevice((int)i_gpu));
        hipStream_t s = sblas_rt::stream(i_gpu);
        if (policy == segment) {
            const size_t first = (order == row_major) ? (size_t)height : (size_t)width;
            const size_t second = (order == row_major) ? (size_t)width : (size_t)height;
            size_t off = 0, dim = 0;
            sblas_detail::dense_block(first, n_gpu, i_gpu, off, dim);
            if (dim)
                CUDA_SAFE_CALL(hipMemcpyAsync(val + off * second, val_gpu[i_gpu], dim * second * sizeof(DataType),
                                              hipMemcpyDeviceToHost, s));
        } else if (policy == replicate) {
            CUDA_SAFE_CALL(hipMemcpyAsync(val, val_gpu[i_gpu], get_mtx_size(), hipMemcpyDeviceToHost, s));
        }
        CUDA_SAFE_CALL(hipStreamSynchronize(s));
    }
    // this = beta*this + alpha*dm on every GPU copy
    void plusDenseMatrixGPU(DenseMatrix const &dm, DataType alpha, DataType beta)
    {
        if (n_gpu == 0 || policy == none) return;
        const size_t cnt = get_mtx_num();
        for (unsigned i = 0; i < n_gpu; ++i) {
            CUDA_SAFE_CALL(cudaSetDevice((int)i));
            if (std::is_same<DataType, double>::value || std::is_same<DataType, float>::value) {
                sblas_rt::must_sblas(sblas_hip_axpby(-1, sblas_rt::stream(i), sblas_rt::vtype_of<DataType>("axpby"),
                                                     (int64_t)cnt, (double)alpha, dm.val_gpu[i], (double)beta, val_gpu[i]),
                                     "sblas_hip_axpby");
            } else {
                const unsigned blocks = (unsigned)std::min<size_t>((cnt + NUM_THREADS_PER_BLK - 1) / NUM_THREADS_PER_BLK, 2048);
                hipLaunchKernelGGL((denseVector_plusEqual_denseVector<size_t, DataType>), dim3(blocks ? blocks : 1),
                                   dim3(NUM_THREADS_PER_BLK), 0, sblas_rt::stream(i), val_gpu[i], dm.val_gpu[i], alpha,
                                   beta, cnt);
            }
        }
        sblas_rt::sync_all(n_gpu);
        CUDA_CHECK_ERROR();
    }
    size_t get_dim_gpu_size(unsigned i_gpu) { return get_dim_gpu_num(i_gpu) * sizeof(DataType); }
    size_t get_dim_gpu_num(unsigned i_gpu)
    {
        assert(i_gpu < n_gpu);
        return (size_t)dim_gpu[i_gpu];
    }
    size_t get_row_size() { return (size_t)width * sizeof(DataType); }
    size_t get_col_size() { return (size_t)height * sizeof(DataType); }
    size_t get_mtx_size() { return (size_t)width * (size_t)height * sizeof(DataType); }
    size_t get_mtx_num() { return (size_t)width * (size_t)height; }

  public:
    IdxType height;
    IdxType width;
    DataType *val;
    DataType **val_gpu;
    IdxType *dim_gpu; // leading-dimension units (columns if col_major) held by each GPU under `segment`
    unsigned n_gpu;
    enum GpuSharePolicy policy;
    enum MajorOrder order;
};

// ----------------------------------------------------------------------------------------------
// Dense vector (replicated on the GPUs; never segmented)
// ----------------------------------------------------------------------------------------------
template <typename IdxType, typename DataType> class DenseVector {
  public:
    DenseVector() : length(0), val(NULL), val_gpu(NULL), n_gpu(0), policy(none) {}
    DenseVector(IdxType _length) : length(_length), val(NULL), val_gpu(NULL), n_gpu(0), policy(none)
    {
        SAFE_ALOC_HOST(val, get_vec_size());
        srand(RAND_INIT_SEED);
        for (size_t i = 0; i < get_vec_length(); ++i) val[i] = (DataType)rand0to1();
    }
    DenseVector(IdxType _length, DataType _val) : length(_length), val(NULL), val_gpu(NULL), n_gpu(0), policy(none)
    {
        SAFE_ALOC_HOST(val, get_vec_size());
        srand(RAND_INIT_SEED);
        for (size_t i = 0; i < get_vec_length(); ++i) val[i] = _val;
    }
    DenseVector(const DenseVector &dv) : length(dv.length), val(NULL), val_gpu(NULL), n_gpu(dv.n_gpu), policy(dv.policy)
    {
        SAFE_ALOC_HOST(val, get_vec_size());
        memcpy(val, dv.val, get_vec_size());
        if (n_gpu != 0 && policy != none) {
            SAFE_ALOC_HOST(val_gpu, n_gpu * sizeof(DataType *));
            for (unsigned i = 0; i < n_gpu; ++i) {
                CUDA_SAFE_CALL(cudaSetDevice((int)i));
                SAFE_ALOC_GPU(val_gpu[i], get_vec_size());
                CUDA_SAFE_CALL(hipMemcpyAsync(val_gpu[i], dv.val_gpu[i], get_vec_size(), hipMemcpyDeviceToDevice,
                                              sblas_rt::stream(i)));
            }
            sblas_rt::sync_all(n_gpu);
        }
    }
    ~DenseVector()
    {
        SAFE_FREE_HOST(val);
        SAFE_FREE_MULTI_GPU(val_gpu, n_gpu);
    }
    void sync2gpu(unsigned _n_gpu, enum GpuSharePolicy _policy)
    {
        n_gpu = _n_gpu;
        policy = _policy;
        assert(n_gpu != 0);
        assert(policy != none);
        assert(policy != segment); // vectors are never partitioned
        SAFE_ALOC_HOST(val_gpu, n_gpu * sizeof(DataType *));
        sblas_detail::replicate_async(n_gpu, val, get_vec_length(), val_gpu);
        sblas_rt::sync_all(n_gpu);
    }
    void sync2cpu(unsigned i_gpu) // every GPU holds the same result; take any
    {
        assert(i_gpu < n_gpu);
        assert(val_gpu != NULL);
        CUDA_SAFE_CALL(cudaSetDevice((int)i_gpu));
        hipStream_t s = sblas_rt::stream(i_gpu);
        CUDA_SAFE_CALL(hipMemcpyAsync(val, val_gpu[i_gpu], get_vec_size(), hipMemcpyDeviceToHost, s));
        CUDA_SAFE_CALL(hipStreamSynchronize(s));
    }
    void plusDenseVectorGPU(DenseVector const &dv, DataType alpha, DataType beta)
    {
        if (n_gpu == 0 || policy == none) return;
        const size_t cnt = get_vec_length();
        for (unsigned i = 0; i < n_gpu; ++i) {
            CUDA_SAFE_CALL(cudaSetDevice((int)i));
            if (std::is_same<DataType, double>::value || std::is_same<DataType, float>::value) {
                sblas_rt::must_sblas(sblas_hip_axpby(-1, sblas_rt::stream(i), sblas_rt::vtype_of<DataType>("axpby"),
                                                     (int64_t)cnt, (double)alpha, dv.val_gpu[i], (double)beta, val_gpu[i]),
                                     "sblas_hip_axpby");
            } else {
                const unsigned blocks = (unsigned)std::min<size_t>((cnt + NUM_THREADS_PER_BLK - 1) / NUM_THREADS_PER_BLK, 2048);
                hipLaunchKernelGGL((denseVector_plusEqual_denseVector<size_t, DataType>), dim3(blocks ? blocks : 1),
                                   dim3(NUM_THREADS_PER_BLK), 0, sblas_rt::stream(i), val_gpu[i], dv.val_gpu[i], alpha,
                                   beta, cnt);
            }
        }
        sblas_rt::sync_all(n_gpu);
        CUDA_CHECK_ERROR();
    }
    size_t get_vec_size() { return (size_t)length * sizeof(DataType); }
    size_t get_vec_length() { return (size_t)length; }

  public:
    IdxType length;
    DataType *val;
    DataType **val_gpu;
    unsigned n_gpu;
    enum GpuSharePolicy policy;
};

#endif
