// comm.hip -- partial-result merge across the GPUs driven by one process.
// RCCL (AMD's NCCL) is dlopen'ed on first use: libsblas_hip.so itself has no link-time dependency
// on it, so the library also loads inside a Python process whose torch wheel ships its own RCCL
// (there the Python side owns the communicator and calls torch.distributed instead).
// Replaces spmm.h:179-181,189,260-262,279 and spmv.h:43-45,58,115-118,134.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdlib.h>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include "../../include/sblas_hip.h"
#include "kernels.h"

namespace {

// Minimal declarations of the RCCL entry points used (rccl.h:236, :260, :611, :919ff).
typedef void *rcclComm_t;
typedef int (*fn_CommInitAll)(rcclComm_t *, int, const int *);
typedef int (*fn_CommDestroy)(rcclComm_t);
typedef int (*fn_AllReduce)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t);
typedef int (*fn_SendRecv)(void *, size_t, int, int, rcclComm_t, hipStream_t); // ncclSend / ncclRecv (rccl.h)
typedef int (*fn_Group)(void);
constexpr int RCCL_FLOAT32 = 7; // ncclFloat32 / ncclFloat, rccl.h:466
constexpr int RCCL_FLOAT64 = 8; // ncclFloat64 / ncclDouble, rccl.h:467
constexpr int RCCL_SUM = 0;     // ncclSum, rccl.h:448

struct Rccl {
    void *handle = nullptr;
    fn_CommInitAll CommInitAll = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_AllReduce AllReduce = nullptr;
    fn_SendRecv Send = nullptr, Recv = nullptr;
    fn_Group GroupStart = nullptr, GroupEnd = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // SBLAS_RCCL_LIB: another library with the same seven entry points (tests/rccl_stub: lets the exchange branches
        // below execute on a one-GPU box; never set in production)
        const char *over = getenv("SBLAS_RCCL_LIB");
        const char *names[] = {over && *over ? over : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle || (over && *over)) break; // (an override that does not load is an error, not a fallback)
        }
        if (!r.handle) return;
        r.CommInitAll = (fn_CommInitAll)dlsym(r.handle, "ncclCommInitAll");
        r.CommDestroy = (fn_CommDestroy)dlsym(r.handle, "ncclCommDestroy");
        r.AllReduce = (fn_AllReduce)dlsym(r.handle, "ncclAllReduce");
        r.Send = (fn_SendRecv)dlsym(r.handle, "ncclSend");
        r.Recv = (fn_SendRecv)dlsym(r.handle, "ncclRecv");
        r.GroupStart = (fn_Group)dlsym(r.handle, "ncclGroupStart");
        r.GroupEnd = (fn_Group)dlsym(r.handle, "ncclGroupEnd");
        r.ok = r.CommInitAll && r.CommDestroy && r.AllReduce && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
    });
    return r;
}

struct CommSet {
    std::vector<int> devs;
    bool one_device = false;         // every rank sits on the same physical device
    std::vector<rcclComm_t> comms;   // empty when one_device or a single rank
    std::vector<hipEvent_t> events;  // per rank, for the one-device path
};

std::mutex g_mu;
std::map<std::vector<int>, std::unique_ptr<CommSet>> g_sets;

} // namespace

extern "C" int sblas_hip_comm_get(int n_gpu, const int *devs, void **comm_out)
{
    if (n_gpu <= 0 || n_gpu > sblas::MAX_REPLICAS || !comm_out) return SBLAS_E_INVALID;
    std::vector<int> key(n_gpu);
    for (int i = 0; i < n_gpu; ++i) key[i] = devs ? devs[i] : i;
    // TEST SWITCH (with SBLAS_RCCL_LIB = the stub of tests/rccl_stub): ranks that share a device take the exchange
    // path -- communicator, grouped send / recv, grouped all-reduce -- instead of the one-device fold.  A real RCCL
    // rejects duplicate devices in ncclCommInitAll, so the switch does nothing useful outside the tests.
    const char *fe = getenv("SBLAS_COMM_FORCE_EXCHANGE");
    const bool force_exchange = fe && *fe && *fe != '0';
    if (force_exchange) key.push_back(-1); // (a cache entry of its own)
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_sets.find(key);
    if (it != g_sets.end()) {
        *comm_out = it->second.get();
        return SBLAS_OK;
    }
    auto set = std::make_unique<CommSet>();
    set->devs.assign(key.begin(), key.begin() + n_gpu);
    bool all_same = true, all_distinct = true;
    for (int i = 0; i < n_gpu; ++i)
        for (int j = i + 1; j < n_gpu; ++j) {
            if (key[i] != key[j]) all_same = false;
            else all_distinct = false;
        }
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = SBLAS_OK;
    if (n_gpu == 1) {
        // a single rank: the sum over ranks is the identity, no communicator needed
    } else if (all_same && !force_exchange) {
        set->one_device = true;
        if (hipSetDevice(key[0]) != hipSuccess) rc = SBLAS_E_HIP;
        for (int i = 0; i < n_gpu && rc == SBLAS_OK; ++i) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = SBLAS_E_HIP;
            else set->events.push_back(e);
        }
    } else if (all_distinct || force_exchange) {
        Rccl &r = rccl();
        if (!r.ok) rc = SBLAS_E_RCCL;
        if (rc == SBLAS_OK) {
            set->comms.assign(n_gpu, nullptr);
            if (r.CommInitAll(set->comms.data(), n_gpu, set->devs.data()) != 0) {
                set->comms.clear();
                rc = SBLAS_E_RCCL;
            }
        }
    } else {
        rc = SBLAS_E_INVALID; // partially oversubscribed layouts are not supported
    }
    (void)hipSetDevice(prev); // every path leaves the caller's device current
    if (rc != SBLAS_OK) {     // nothing half-built stays behind
        for (auto e : set->events) (void)hipEventDestroy(e);
        return rc;
    }
    *comm_out = set.get();
    g_sets.emplace(key, std::move(set));
    return SBLAS_OK;
}

extern "C" void sblas_hip_comm_release_all(void)
{
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto &kv : g_sets) {
        CommSet &s = *kv.second;
        for (auto c : s.comms) rccl().CommDestroy(c);
        for (auto e : s.events) (void)hipEventDestroy(e);
    }
    g_sets.clear();
}

extern "C" int sblas_hip_allreduce_sum(void *comm, int vtype, void *const *bufs, void *const *streams, int64_t count)
{
    if (vtype != SBLAS_F64 && vtype != SBLAS_F32) return SBLAS_E_INVALID;
    if (!comm || !bufs || count < 0) return SBLAS_E_INVALID;
    if (count == 0) return SBLAS_OK;
    CommSet &s = *static_cast<CommSet *>(comm);
    const int g = (int)s.devs.size();
    for (int i = 0; i < g; ++i)
        if (!bufs[i]) return SBLAS_E_INVALID;
    if (g == 1) return SBLAS_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = SBLAS_OK;
    if (s.one_device) {
        // all ranks share a device: order rank 0's stream after every rank's producers, sum there,
        // then order every other stream after the sum.
        if (hipSetDevice(s.devs[0]) != hipSuccess) return SBLAS_E_HIP;
        hipStream_t s0 = streams ? (hipStream_t)streams[0] : nullptr;
        for (int i = 1; i < g && rc == SBLAS_OK; ++i) {
            hipStream_t si = streams ? (hipStream_t)streams[i] : nullptr;
            if (si == s0) continue;
            if (hipEventRecord(s.events[i], si) != hipSuccess || hipStreamWaitEvent(s0, s.events[i], 0) != hipSuccess)
                rc = SBLAS_E_HIP;
        }
        if (rc == SBLAS_OK && sblas::launch_typed_sum_replicas(s0, vtype, bufs, g, count) != hipSuccess) rc = SBLAS_E_HIP;
        if (rc == SBLAS_OK && hipEventRecord(s.events[0], s0) != hipSuccess) rc = SBLAS_E_HIP;
        for (int i = 1; i < g && rc == SBLAS_OK; ++i) {
            hipStream_t si = streams ? (hipStream_t)streams[i] : nullptr;
            if (si == s0) continue;
            if (hipStreamWaitEvent(si, s.events[0], 0) != hipSuccess) rc = SBLAS_E_HIP;
        }
    } else {
        Rccl &r = rccl();
        if (!r.ok) return SBLAS_E_RCCL;
        if (r.GroupStart() != 0) return SBLAS_E_RCCL;
        for (int i = 0; i < g; ++i) {
            if (hipSetDevice(s.devs[i]) != hipSuccess) { rc = SBLAS_E_HIP; break; }
            if (r.AllReduce(bufs[i], bufs[i], (size_t)count, vtype == SBLAS_F32 ? RCCL_FLOAT32 : RCCL_FLOAT64, RCCL_SUM, s.comms[i],
                            streams ? (hipStream_t)streams[i] : nullptr) != 0) { rc = SBLAS_E_RCCL; break; }
        }
        if (r.GroupEnd() != 0 && rc == SBLAS_OK) rc = SBLAS_E_RCCL;
    }
    (void)hipSetDevice(prev);
    return rc;
}

extern "C" int sblas_hip_allreduce_sum_f64(void *comm, double *const *bufs, void *const *streams, int64_t count)
{
    return sblas_hip_allreduce_sum(comm, SBLAS_F64, reinterpret_cast<void *const *>(bufs), streams, count);
}

extern "C" int sblas_hip_merge_rowblocks_local_f64(int device, void *stream, int64_t M, int64_t N, int g,
                                                   const int64_t *start_row, const int64_t *num_rows,
                                                   const double *const *src, double alpha, double beta, double *C,
                                                   int64_t ldc)
{
    if (M < 0 || N < 0 || g <= 0 || g > sblas::MAX_REPLICAS || !start_row || !num_rows || !src || ldc < M) return SBLAS_E_INVALID;
    if (M == 0 || N == 0) return SBLAS_OK;
    if (!C) return SBLAS_E_INVALID;
    for (int q = 0; q < g; ++q) {
        if (num_rows[q] < 0 || start_row[q] < 0 || start_row[q] + num_rows[q] > M) return SBLAS_E_INVALID;
        if (num_rows[q] > 0 && !src[q]) return SBLAS_E_INVALID;
    }
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return SBLAS_E_HIP;
    return sblas::launch_merge_rowblocks((hipStream_t)stream, M, N, g, src, start_row, num_rows, alpha, beta, C, ldc) ==
                   hipSuccess
               ? SBLAS_OK
               : SBLAS_E_HIP;
}

extern "C" int sblas_hip_merge_rowblocks(void *comm, int vtype, int64_t M, int64_t N, const int64_t *start_row,
                                         const int64_t *num_rows, void *const *partial, void *const *gather, double alpha,
                                         double beta, void *const *C, int64_t ldc, void *const *streams)
{
    if (vtype != SBLAS_F64 && vtype != SBLAS_F32) return SBLAS_E_INVALID;
    const size_t esz = vtype == SBLAS_F32 ? 4 : 8;
    const int rccl_type = vtype == SBLAS_F32 ? RCCL_FLOAT32 : RCCL_FLOAT64;
    if (!comm || !start_row || !num_rows || !partial || !C || M < 0 || N < 0 || ldc < M) return SBLAS_E_INVALID;
    if (M == 0 || N == 0) return SBLAS_OK;
    CommSet &s = *static_cast<CommSet *>(comm);
    const int g = (int)s.devs.size();
    for (int q = 0; q < g; ++q) {
        if (num_rows[q] < 0 || start_row[q] < 0 || start_row[q] + num_rows[q] > M || !C[q]) return SBLAS_E_INVALID;
        if (num_rows[q] > 0 && !partial[q]) return SBLAS_E_INVALID;
    }
    const bool exchange = g > 1 && !s.one_device;
    if (exchange && !gather) return SBLAS_E_INVALID;
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = SBLAS_OK;
    std::vector<size_t> off(g + 1, 0); // packed blocks back to back in every gather buffer
    for (int q = 0; q < g; ++q) off[q + 1] = off[q] + (size_t)num_rows[q] * (size_t)N;
    if (s.one_device && g > 1) {
        // all ranks on one device: every rank's merge must see every rank's block -> order each stream after all
        if (hipSetDevice(s.devs[0]) != hipSuccess) return SBLAS_E_HIP;
        for (int i = 0; i < g && rc == SBLAS_OK; ++i)
            if (hipEventRecord(s.events[i], streams ? (hipStream_t)streams[i] : nullptr) != hipSuccess) rc = SBLAS_E_HIP;
        for (int i = 0; i < g && rc == SBLAS_OK; ++i)
            for (int q = 0; q < g && rc == SBLAS_OK; ++q)
                if (q != i && hipStreamWaitEvent(streams ? (hipStream_t)streams[i] : nullptr, s.events[q], 0) != hipSuccess)
                    rc = SBLAS_E_HIP;
    } else if (exchange) {
        Rccl &r = rccl();
        if (!r.ok) return SBLAS_E_RCCL;
        for (int i = 0; i < g; ++i)
            if (!gather[i]) return SBLAS_E_INVALID;
        // all-to-all of the packed blocks: point-to-point transfers over the xGMI mesh, one group
        if (r.GroupStart() != 0) return SBLAS_E_RCCL;
        for (int i = 0; i < g && rc == SBLAS_OK; ++i) {
            if (hipSetDevice(s.devs[i]) != hipSuccess) { rc = SBLAS_E_HIP; break; }
            hipStream_t si = streams ? (hipStream_t)streams[i] : nullptr;
            for (int q = 0; q < g && rc == SBLAS_OK; ++q) {
                if (q == i) continue;
                const size_t mine = (size_t)num_rows[i] * (size_t)N, theirs = (size_t)num_rows[q] * (size_t)N;
                if (mine && r.Send(partial[i], mine, rccl_type, q, s.comms[i], si) != 0) rc = SBLAS_E_RCCL;
                if (theirs && rc == SBLAS_OK &&
                    r.Recv(static_cast<char *>(gather[i]) + off[q] * esz, theirs, rccl_type, q, s.comms[i], si) != 0)
                    rc = SBLAS_E_RCCL;
            }
        }
        if (r.GroupEnd() != 0 && rc == SBLAS_OK) rc = SBLAS_E_RCCL;
    }
    for (int i = 0; i < g && rc == SBLAS_OK; ++i) {
        if (hipSetDevice(s.devs[i]) != hipSuccess) { rc = SBLAS_E_HIP; break; }
        const void *src[sblas::MAX_REPLICAS];
        for (int q = 0; q < g; ++q)
            src[q] = (q == i || !exchange) ? partial[q] : static_cast<const void *>(static_cast<char *>(gather[i]) + off[q] * esz);
        hipStream_t si = streams ? (hipStream_t)streams[i] : nullptr;
        const hipError_t e =
            vtype == SBLAS_F64
                ? sblas::launch_merge_rowblocks(si, M, N, g, reinterpret_cast<const double *const *>(src), start_row, num_rows,
                                                alpha, beta, static_cast<double *>(C[i]), ldc)
                : sblas::launch_typed_merge_rowblocks(si, vtype, M, N, g, src, start_row, num_rows, alpha, beta, C[i], ldc);
        if (e != hipSuccess) rc = SBLAS_E_HIP;
    }
    (void)hipSetDevice(prev);
    return rc;
}

extern "C" int sblas_hip_merge_rowblocks_f64(void *comm, int64_t M, int64_t N, const int64_t *start_row,
                                             const int64_t *num_rows, double *const *partial, double *const *gather,
                                             double alpha, double beta, double *const *C, int64_t ldc,
                                             void *const *streams)
{
    return sblas_hip_merge_rowblocks(comm, SBLAS_F64, M, N, start_row, num_rows, reinterpret_cast<void *const *>(partial),
                                     reinterpret_cast<void *const *>(gather), alpha, beta,
                                     reinterpret_cast<void *const *>(C), ldc, streams);
}
