// rccl_stub.cpp -- TEST INFRASTRUCTURE, not part of the product.
//
// A stand-in for librccl with the seven entry points comm.hip resolves (ncclCommInitAll, ncclCommDestroy,
// ncclAllReduce, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd), so that the exchange branches of comm.hip --
// grouped send / recv of the packed row blocks with their gather offsets, grouped all-reduce -- can execute on a box
// with ONE GPU (the pool's boxes): SBLAS_RCCL_LIB points comm.hip at this library and SBLAS_COMM_FORCE_EXCHANGE=1 lets
// equal device ids take the distinct-device path.  What it checks on the way, because a real RCCL would hang or
// corrupt memory where this returns an error:
//   * every ncclSend(i -> q, count, type) of a group has exactly one matching ncclRecv(q <- i) with the same count and
//     type, and vice versa (ncclGroupEnd returns ncclInvalidUsage otherwise);
//   * data types are the enum values of rccl.h (7 = ncclFloat32, 8 = ncclFloat64), the reduction is ncclSum (0);
//   * an all-reduce is posted by every rank of the communicator in one group, in place, with equal counts.
// Transfers are stream-ordered like the real thing: the receiver's stream waits for the sender's stream, copies, and the
// sender's stream waits for the copy (so the send buffer may be reused afterwards, as NCCL's stream semantics promise).
// The all-reduce is a host round trip (rank-ordered sum): slow and synchronous, but exact and obviously right.
// What stays unverified: the real RCCL (hand-declared prototypes against the real ABI), xGMI.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <vector>

namespace {
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };

struct World {
    int n = 0;
    std::vector<int> devs;
};
struct Comm {
    World *world;
    int rank;
};
struct Op {
    enum Kind { SEND, RECV, ALLREDUCE } kind;
    Comm *comm;
    void *buf;
    size_t count;
    int dtype;
    int peer;
    hipStream_t stream;
    bool done = false;
};
std::mutex g_mu;
int g_depth = 0;
std::vector<Op> g_ops;
long long g_stats[4] = {0, 0, 0, 0}; // sends matched, all-reduces carried out, bytes sent, groups

size_t type_size(int dtype) { return dtype == 7 ? 4 : dtype == 8 ? 8 : 0; }

int run_group()
{
    ++g_stats[3];
    int rc = ncclSuccess;
    // point-to-point: every send meets its receive (FIFO per ordered pair)
    for (size_t a = 0; a < g_ops.size() && rc == ncclSuccess; ++a) {
        Op &s = g_ops[a];
        if (s.kind != Op::SEND || s.done) continue;
        Op *r = nullptr;
        for (size_t b = 0; b < g_ops.size(); ++b) {
            Op &c = g_ops[b];
            if (c.kind == Op::RECV && !c.done && c.comm->world == s.comm->world && c.comm->rank == s.peer &&
                c.peer == s.comm->rank) {
                r = &c;
                break;
            }
        }
        if (!r || r->count != s.count || r->dtype != s.dtype) {
            fprintf(stderr, "rccl_stub: send %d -> %d (%zu x type %d) has no matching receive\n", s.comm->rank, s.peer, s.count, s.dtype);
            rc = ncclInvalidUsage;
            break;
        }
        const size_t bytes = s.count * type_size(s.dtype);
        hipEvent_t ready = nullptr, copied = nullptr;
        const int sdev = s.comm->world->devs[s.comm->rank], rdev = r->comm->world->devs[r->comm->rank];
        if (hipSetDevice(sdev) != hipSuccess || hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(ready, s.stream) != hipSuccess)
            rc = ncclUnhandledCudaError;
        if (rc == ncclSuccess &&
            (hipSetDevice(rdev) != hipSuccess || hipStreamWaitEvent(r->stream, ready, 0) != hipSuccess ||
             hipMemcpyAsync(r->buf, s.buf, bytes, hipMemcpyDeviceToDevice, r->stream) != hipSuccess ||
             hipEventCreateWithFlags(&copied, hipEventDisableTiming) != hipSuccess || hipEventRecord(copied, r->stream) != hipSuccess))
            rc = ncclUnhandledCudaError;
        if (rc == ncclSuccess && (hipSetDevice(sdev) != hipSuccess || hipStreamWaitEvent(s.stream, copied, 0) != hipSuccess))
            rc = ncclUnhandledCudaError;
        if (ready) (void)hipEventDestroy(ready);   // (destruction is deferred until the recorded work has completed)
        if (copied) (void)hipEventDestroy(copied);
        s.done = r->done = true;
        ++g_stats[0];
        g_stats[2] += (long long)bytes;
    }
    for (Op &o : g_ops)
        if (rc == ncclSuccess && o.kind == Op::RECV && !o.done) {
            fprintf(stderr, "rccl_stub: receive %d <- %d (%zu x type %d) has no matching send\n", o.comm->rank, o.peer, o.count, o.dtype);
            rc = ncclInvalidUsage;
        }
    // all-reduce: one entry per rank of the world, equal counts and types; rank-ordered sum through the host
    for (size_t a = 0; a < g_ops.size() && rc == ncclSuccess; ++a) {
        Op &first = g_ops[a];
        if (first.kind != Op::ALLREDUCE || first.done) continue;
        World *w = first.comm->world;
        std::vector<Op *> per(w->n, nullptr);
        for (Op &o : g_ops)
            if (o.kind == Op::ALLREDUCE && !o.done && o.comm->world == w && !per[o.comm->rank]) per[o.comm->rank] = &o;
        for (int i = 0; i < w->n; ++i)
            if (!per[i] || per[i]->count != first.count || per[i]->dtype != first.dtype) {
                fprintf(stderr, "rccl_stub: all-reduce not posted alike by every rank (rank %d)\n", i);
                rc = ncclInvalidUsage;
            }
        if (rc != ncclSuccess) break;
        const size_t bytes = first.count * type_size(first.dtype);
        std::vector<std::vector<char>> host(w->n, std::vector<char>(bytes));
        for (int i = 0; i < w->n && rc == ncclSuccess; ++i)
            if (hipSetDevice(w->devs[i]) != hipSuccess || hipStreamSynchronize(per[i]->stream) != hipSuccess ||
                hipMemcpy(host[i].data(), per[i]->buf, bytes, hipMemcpyDeviceToHost) != hipSuccess)
                rc = ncclUnhandledCudaError;
        if (rc != ncclSuccess) break;
        std::vector<char> sum(bytes);
        if (first.dtype == 8) {
            double *d = reinterpret_cast<double *>(sum.data());
            for (size_t e = 0; e < first.count; ++e) {
                double t = 0.0;
                for (int i = 0; i < w->n; ++i) t += reinterpret_cast<const double *>(host[i].data())[e];
                d[e] = t;
            }
        } else {
            float *d = reinterpret_cast<float *>(sum.data());
            for (size_t e = 0; e < first.count; ++e) {
                float t = 0.0f;
                for (int i = 0; i < w->n; ++i) t += reinterpret_cast<const float *>(host[i].data())[e];
                d[e] = t;
            }
        }
        for (int i = 0; i < w->n && rc == ncclSuccess; ++i) {
            if (hipSetDevice(w->devs[i]) != hipSuccess || hipMemcpy(per[i]->buf, sum.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
                rc = ncclUnhandledCudaError;
            per[i]->done = true;
        }
        ++g_stats[1];
    }
    g_ops.clear();
    return rc;
}
} // namespace

extern "C" {
int ncclCommInitAll(void **comms, int ndev, const int *devlist)
{
    if (!comms || ndev <= 0) return ncclInvalidArgument;
    World *w = new World;
    w->n = ndev;
    for (int i = 0; i < ndev; ++i) w->devs.push_back(devlist ? devlist[i] : i);
    for (int i = 0; i < ndev; ++i) comms[i] = new Comm{w, i};
    return ncclSuccess;
}
int ncclCommDestroy(void *comm)
{
    delete static_cast<Comm *>(comm); // (the World is leaked: a handful of bytes per test process)
    return ncclSuccess;
}
int ncclGroupStart(void)
{
    std::lock_guard<std::mutex> lock(g_mu);
    ++g_depth;
    return ncclSuccess;
}
int ncclGroupEnd(void)
{
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    int prev = 0;
    (void)hipGetDevice(&prev);
    const int rc = run_group();
    (void)hipSetDevice(prev);
    return rc;
}
static int post(Op::Kind kind, void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    if (!comm || !buf || type_size(dtype) == 0) return ncclInvalidArgument;
    Comm *c = static_cast<Comm *>(comm);
    if (kind != Op::ALLREDUCE && (peer < 0 || peer >= c->world->n || peer == c->rank)) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lock(g_mu);
    Op o;
    o.kind = kind, o.comm = c, o.buf = buf, o.count = count, o.dtype = dtype, o.peer = peer, o.stream = stream;
    g_ops.push_back(o);
    if (g_depth > 0) return ncclSuccess;
    // outside a group a lone point-to-point call would block for ever in the real library: refuse
    int prev = 0;
    (void)hipGetDevice(&prev);
    const int rc = run_group();
    (void)hipSetDevice(prev);
    return rc;
}
int ncclSend(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    return post(Op::SEND, buf, count, dtype, peer, comm, stream);
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    return post(Op::RECV, buf, count, dtype, peer, comm, stream);
}
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t stream)
{
    if (send != recv || op != 0) return ncclInvalidArgument; // comm.hip reduces in place with ncclSum
    return post(Op::ALLREDUCE, recv, count, dtype, -1, comm, stream);
}
// test hook: [0] sends matched, [1] all-reduces, [2] bytes sent, [3] groups
void rccl_stub_stats(long long out[4], int reset)
{
    std::lock_guard<std::mutex> lock(g_mu);
    for (int i = 0; i < 4; ++i) {
        out[i] = g_stats[i];
        if (reset) g_stats[i] = 0;
    }
}
}
