// utility.h -- runtime utilities of the S-BLAS API layer on HIP.
//
// Keeps the names the reference's drivers use (reference utility.h:27-193, :197, :276-300):
// CUDA_SAFE_CALL / CUDA_CHECK_ERROR, SAFE_ALOC_* / SAFE_FREE_*, cpu_timer / gpu_timer, check_equal, rand0to1,
// exclusive_scan, csr_findRowIdxUsingNnzIdx, print_1d_array -- plus the few CUDA runtime spellings that appear
// literally in driver code (cudaDeviceSynchronize ...), mapped onto HIP.  Error convention of the reference
// is kept at this level: print and exit(-1).
#ifndef SBLAS_AMD_UTILITY_H
#define SBLAS_AMD_UTILITY_H

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <sys/time.h>
#include <type_traits>
#include <vector>

#include "config.h"
#include "sblas_hip.h"

// ----------------------------------------------------------------------------------------------
// logical GPU i of the S-BLAS API -> physical HIP device, streams, workspaces
// ----------------------------------------------------------------------------------------------
namespace sblas_rt {

inline int physical_device_count()
{
    static const int n = sblas_hip_device_count();
    return n;
}
inline bool have_gpu() { return physical_device_count() > 0; }

// The reference addresses devices 0..n_gpu-1 directly (spmm.h:101-104).  When a driver asks for more GPUs than
// the node has (unit_test.cu hard-codes 4), logical GPUs are folded onto the physical ones so that the g-way
// placement and merge logic can still be exercised; the merge then needs all ranks on ONE device (see
// sblas_hip_comm_get), i.e. folding is only supported onto a single-GPU box.
inline int physical_device(unsigned logical)
{
    const int n = physical_device_count();
    if (n <= 0) {
        fprintf(stderr, "S-BLAS: no HIP device visible; the GPU paths have no CPU fallback\n");
        exit(-1);
    }
    static bool told = false;
    if ((int)logical >= n && !told) {
        told = true;
        fprintf(stderr, "S-BLAS: %u logical GPUs requested, %d present: folding logical GPUs onto physical ones\n",
                logical + 1, n);
    }
    return (int)(logical % (unsigned)n);
}

// scratch slots per logical GPU (all grow-only, kept across op calls)
enum { WS_STAGING = 0, WS_PARTIAL = 1, WS_GATHER = 2, WS_SLOTS = 3 };
struct PerGpu {
    hipStream_t stream = nullptr;
    hipStream_t merge_stream = nullptr;   // method 2: the exchange + scatter of column tile c runs here, beside the SpMM of tile c + 1
    std::vector<hipEvent_t> events;       // (grow-only pool)
    void *workspace[WS_SLOTS] = {nullptr, nullptr, nullptr};
    size_t workspace_bytes[WS_SLOTS] = {0, 0, 0};
};
inline std::map<unsigned, PerGpu> &table()
{
    static std::map<unsigned, PerGpu> t;
    return t;
}
inline std::mutex &table_mutex()
{
    static std::mutex m;
    return m;
}

inline void must(hipError_t e, const char *what)
{
    if (e != hipSuccess) {
        fprintf(stderr, "S-BLAS: %s failed: %s\n", what, hipGetErrorString(e));
        exit(-1);
    }
}
inline void must_sblas(int rc, const char *what)
{
    if (rc != SBLAS_OK) {
        fprintf(stderr, "S-BLAS: %s failed: %s\n", what, sblas_hip_error_string(rc));
        exit(-1);
    }
}

// one non-blocking stream per logical GPU, created on first use
inline hipStream_t stream(unsigned logical)
{
    std::lock_guard<std::mutex> lock(table_mutex());
    PerGpu &g = table()[logical];
    if (!g.stream) {
        must(hipSetDevice(physical_device(logical)), "hipSetDevice");
        must(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking), "hipStreamCreate");
    }
    return g.stream;
}

// the second stream of a logical GPU (method 2's merge) and events to order it against the first
inline hipStream_t merge_stream(unsigned logical)
{
    std::lock_guard<std::mutex> lock(table_mutex());
    PerGpu &g = table()[logical];
    if (!g.merge_stream) {
        must(hipSetDevice(physical_device(logical)), "hipSetDevice");
        must(hipStreamCreateWithFlags(&g.merge_stream, hipStreamNonBlocking), "hipStreamCreate");
    }
    return g.merge_stream;
}
inline hipEvent_t event(unsigned logical, size_t k)
{
    std::lock_guard<std::mutex> lock(table_mutex());
    PerGpu &g = table()[logical];
    while (g.events.size() <= k) {
        hipEvent_t e = nullptr;
        must(hipSetDevice(physical_device(logical)), "hipSetDevice");
        must(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
        g.events.push_back(e);
    }
    return g.events[k];
}

// grow-only scratch buffers per logical GPU: the "externalBuffer" of the replaced cuSPARSE calls (spmm.h:134-141)
// and the partial-result / gather buffers of method 2 (the reference builds a DenseMatrix C_copy and uploads M*N
// zeros inside every call, spmm.h:182-183), kept instead of being malloc'ed and freed inside every op call
inline void *workspace(unsigned logical, size_t bytes, int slot = WS_STAGING)
{
    std::lock_guard<std::mutex> lock(table_mutex());
    PerGpu &g = table()[logical];
    if (bytes == 0) bytes = 16;
    if (g.workspace_bytes[slot] < bytes) {
        must(hipSetDevice(physical_device(logical)), "hipSetDevice");
        if (g.workspace[slot]) must(hipFree(g.workspace[slot]), "hipFree"); // (hipFree drains the device first)
        must(hipMalloc(&g.workspace[slot], bytes), "hipMalloc(workspace)");
        g.workspace_bytes[slot] = bytes;
    }
    return g.workspace[slot];
}

inline void sync_all(unsigned n_gpu)
{
    for (unsigned i = 0; i < n_gpu; ++i) {
        must(hipSetDevice(physical_device(i)), "hipSetDevice");
        must(hipStreamSynchronize(stream(i)), "hipStreamSynchronize");
    }
}

// host allocations: pinned when a device exists (async H2D/D2H), plain otherwise (CPU-only drivers, gpus = 0)
inline void *host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0) bytes = 1;
    if (have_gpu()) must(hipHostMalloc(&p, bytes, hipHostMallocDefault), "hipHostMalloc");
    else {
        p = malloc(bytes);
        if (!p) {
            fprintf(stderr, "S-BLAS: out of host memory\n");
            exit(-1);
        }
    }
    return p;
}
inline void host_free(void *p)
{
    if (!p) return;
    if (have_gpu()) must(hipHostFree(p), "hipHostFree");
    else free(p);
}

// Value / index type tags of the typed C-ABI entry points (the reference's getCudaDataType<T>() /
// getCusparseIndexType<T>(), utility.h:302-316: float, double; int32_t, int64_t).  Anything else: print and exit.
template <typename DataType> inline int vtype_of(const char *who)
{
    if (std::is_same<DataType, double>::value) return SBLAS_F64;
    if (std::is_same<DataType, float>::value) return SBLAS_F32;
    fprintf(stderr, "%s: values must be float or double\n", who);
    exit(-1);
}
template <typename IdxType> inline int itype_of(const char *who)
{
    if (std::is_integral<IdxType>::value && std::is_signed<IdxType>::value && sizeof(IdxType) == 4) return SBLAS_I32;
    if (std::is_integral<IdxType>::value && std::is_signed<IdxType>::value && sizeof(IdxType) == 8) return SBLAS_I64;
    fprintf(stderr, "%s: indices must be 32- or 64-bit signed integers\n", who);
    exit(-1);
}

} // namespace sblas_rt

// ----------------------------------------------------------------------------------------------
// CUDA spellings used literally by driver code written against the reference
// ----------------------------------------------------------------------------------------------
typedef hipError_t cudaError;
typedef hipError_t cudaError_t;
#define cudaSuccess hipSuccess
inline hipError_t cudaDeviceSynchronize()
{
    // a driver's cudaDeviceSynchronize() means "everything I launched is done": drain every device in use
    if (!sblas_rt::have_gpu()) return hipSuccess;
    int prev = 0;
    (void)hipGetDevice(&prev);
    hipError_t rc = hipSuccess;
    for (int d = 0; d < sblas_rt::physical_device_count(); ++d) {
        if (hipSetDevice(d) != hipSuccess) continue;
        const hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) rc = e;
    }
    (void)hipSetDevice(prev);
    return rc;
}
inline hipError_t cudaThreadSynchronize() { return cudaDeviceSynchronize(); }
inline hipError_t cudaSetDevice(int logical) { return hipSetDevice(sblas_rt::physical_device((unsigned)logical)); }
inline hipError_t cudaGetDevice(int *d) { return hipGetDevice(d); }
inline hipError_t cudaGetLastError()
{
    return sblas_rt::have_gpu() ? hipGetLastError() : hipSuccess;
}
inline const char *cudaGetErrorString(hipError_t e) { return hipGetErrorString(e); }

// ----------------------------------------------------------------------------------------------
// error checking (reference utility.h:27-59)
// ----------------------------------------------------------------------------------------------
#define CUDA_SAFE_CALL(err) sblas_safe_call((err), __FILE__, __LINE__)
inline void sblas_safe_call(hipError_t err, const char *file, const int line)
{
#ifdef CUDA_ERROR_CHECK
    if (err != hipSuccess) {
        fprintf(stderr, "cudaSafeCall() failed at %s:%i : %s\n", file, line, hipGetErrorString(err));
        exit(-1);
    }
#endif
}
#define CUDA_CHECK_ERROR() sblas_check_error(__FILE__, __LINE__)
inline void sblas_check_error(const char *file, const int line)
{
#ifdef CUDA_ERROR_CHECK
    hipError_t err = cudaGetLastError();
    if (err != hipSuccess) {
        fprintf(stderr, "cudaCheckError() failed at %s:%i : %s\n", file, line, hipGetErrorString(err));
        exit(-1);
    }
    err = cudaDeviceSynchronize();
    if (err != hipSuccess) {
        fprintf(stderr, "cudaCheckError() with sync failed at %s:%i : %s\n", file, line, hipGetErrorString(err));
        exit(-1);
    }
#endif
}

// ----------------------------------------------------------------------------------------------
// allocation macros (reference utility.h:86-127)
// ----------------------------------------------------------------------------------------------
#define SAFE_ALOC_HOST(X, Y) (X) = static_cast<std::remove_reference_t<decltype(X)>>(sblas_rt::host_alloc((Y)));
#define SAFE_FREE_HOST(X)                                                                                            \
    if ((X) != NULL) {                                                                                               \
        sblas_rt::host_free((X));                                                                                    \
        (X) = NULL;                                                                                                  \
    }
#define SAFE_ALOC_GPU(X, Y) CUDA_SAFE_CALL(hipMalloc((void **)&(X), (Y) ? (Y) : 1));
#define SAFE_FREE_GPU(X)                                                                                             \
    if ((X) != NULL) {                                                                                               \
        CUDA_SAFE_CALL(hipFree((X)));                                                                                \
        (X) = NULL;                                                                                                  \
    }
// X: host table of per-GPU device pointers, Y: number of logical GPUs
#define SAFE_FREE_MULTI_GPU(X, Y)                                                                                    \
    if ((X) != NULL) {                                                                                               \
        int sblas_prev_dev = 0;                                                                                      \
        CUDA_SAFE_CALL(hipGetDevice(&sblas_prev_dev));                                                               \
        for (unsigned sblas_i = 0; sblas_i < (Y); sblas_i++)                                                         \
            if (((X)[sblas_i]) != NULL) {                                                                            \
                CUDA_SAFE_CALL(cudaSetDevice(sblas_i));                                                              \
                CUDA_SAFE_CALL(hipFree((X)[sblas_i]));                                                               \
            }                                                                                                        \
        sblas_rt::host_free((X));                                                                                    \
        (X) = NULL;                                                                                                  \
        CUDA_SAFE_CALL(hipSetDevice(sblas_prev_dev));                                                                \
    }

// ----------------------------------------------------------------------------------------------
// printing, timing, comparison, random, scan, search
// ----------------------------------------------------------------------------------------------
template <typename T> void print_1d_array(T *input, int length)
{
    for (int i = 0; i < length; i++) {
        printf("%.3lf, ", (double)input[i]);
        if ((i + 1) % 10 == 0) printf("\n");
    }
    printf("\n");
}

inline double get_cpu_timer()
{
    struct timeval tp;
    gettimeofday(&tp, NULL);
    return (double)tp.tv_sec * 1e3 + (double)tp.tv_usec * 1e-3; // milliseconds
}

typedef struct CPU_Timer {
    CPU_Timer() : start(0.0), stop(0.0) {}
    void start_timer() { start = get_cpu_timer(); }
    void stop_timer() { stop = get_cpu_timer(); }
    double measure() { return stop - start; }
    double start, stop;
} cpu_timer;

// event timer on a given stream of the current device (reference utility.h:163-178 used stream 0)
typedef struct GPU_Timer {
    explicit GPU_Timer(hipStream_t s = nullptr) : stream(s)
    {
        CUDA_SAFE_CALL(hipEventCreate(&start));
        CUDA_SAFE_CALL(hipEventCreate(&stop));
    }
    ~GPU_Timer()
    {
        (void)hipEventDestroy(start);
        (void)hipEventDestroy(stop);
    }
    void start_timer() { CUDA_SAFE_CALL(hipEventRecord(start, stream)); }
    void stop_timer() { CUDA_SAFE_CALL(hipEventRecord(stop, stream)); }
    double measure()
    {
        CUDA_SAFE_CALL(hipEventSynchronize(stop));
        float ms = 0;
        CUDA_SAFE_CALL(hipEventElapsedTime(&ms, start, stop));
        return (double)ms;
    }
    hipEvent_t start, stop;
    hipStream_t stream;
} gpu_timer;

// true when every |x[i] - y[i]| <= ERROR_BAR (absolute; reference utility.h:182-191) -- size_t loop index
template <typename T> bool check_equal(const T *x, const T *y, size_t m)
{
    bool correct = true;
    for (size_t i = 0; i < m; i++) {
        const double d = (double)x[i] - (double)y[i];
        if (!(std::fabs(d) <= ERROR_BAR)) correct = false; // also false for NaN
    }
    return correct;
}

inline double rand0to1() { return ((double)rand() / (double)RAND_MAX); }

// in-place exclusive prefix sum over `length` entries
template <typename IdxType, typename DataType> void exclusive_scan(DataType *input, IdxType length)
{
    DataType run = 0;
    for (IdxType i = 0; i < length; i++) {
        const DataType here = input[i];
        input[i] = run;
        run += here;
    }
}

// first row r with rowPtr[r] <= nnzIdx < rowPtr[r+1] (empty rows skipped); -1 if none.  O(log M).
template <typename IdxType>
IdxType csr_findRowIdxUsingNnzIdx(const IdxType *rowPtr, IdxType height, IdxType nnzIdx)
{
    if (height <= 0 || nnzIdx < rowPtr[0] || nnzIdx >= rowPtr[height]) return (IdxType)-1;
    const IdxType *hi = std::upper_bound(rowPtr, rowPtr + height + 1, nnzIdx);
    return (IdxType)(hi - rowPtr) - 1;
}

#endif
