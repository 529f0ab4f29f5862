// spmv_kernels.hip -- hand-written gfx950 (wave64) CSR SpMV kernels  y = alpha*A*x + beta*y  (replaces cusparseSpMV,
// spmv.h:104-106).  The launcher at the end picks the kernel by row-length class.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "kernels.h"

namespace sblas {

constexpr int WAVE = 64;
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---------------------------------------------------------------------------------------------
// SpMV: LPR lanes per row (a power of two, 4..64), 256/LPR rows per workgroup.  The lanes of a
// group stride through the row's nonzeros (coalesced col_idx / val streams, x gathered through
// L2), then the partial sums are folded with xor-shuffles inside the wave -- the wave64 successor of
// the reference's unused sum_32_shfl (utility.h:241-246).
// ---------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void spmv_csr_kernel(int rows, const int *__restrict__ rowptr,
                                                      const int *__restrict__ colidx,
                                                      const double *__restrict__ val,
                                                      const double *__restrict__ x, double alpha, double beta,
                                                      double *__restrict__ y)
{
    constexpr int ROWS_PER_BLOCK = 256 / LPR;
    // a row far longer than the lane group was sized for (skewed matrices: the kernel is picked by the AVERAGE row
    // length) would keep its LPR lanes busy for thousands of trips while the chip idles: such rows are set aside and
    // summed by the whole block afterwards, 256 lanes striding through them
    constexpr int LONG = 64 * LPR;
    __shared__ int long_rows[ROWS_PER_BLOCK];
    __shared__ int n_long;
    __shared__ double wsum[4];
    if (threadIdx.x == 0) n_long = 0;
    __syncthreads();
    const int l = threadIdx.x % LPR;
    const int row = blockIdx.x * ROWS_PER_BLOCK + threadIdx.x / LPR;
    double s0 = 0.0, s1 = 0.0;
    bool deferred = false;
    if (row < rows) {
        const int p1 = rowptr[row + 1];
        int p = rowptr[row];
        deferred = p1 - p > LONG;
        if (deferred) {
            if (l == 0) long_rows[atomicAdd(&n_long, 1)] = row;
            p = p1;
        }
        p += l;
        // two slices per trip (four gave fewer resident waves and ran 20 % slower)
        for (; p + LPR < p1; p += 2 * LPR) {
            const int c0 = colidx[p], c1 = colidx[p + LPR];
            const double a0 = val[p], a1 = val[p + LPR];
            s0 = fma(a0, x[c0], s0);
            s1 = fma(a1, x[c1], s1);
        }
        if (p < p1) s0 = fma(val[p], x[colidx[p]], s0);
    }
    double s = s0 + s1;
#pragma unroll
    for (int m = LPR / 2; m > 0; m >>= 1) s += __shfl_xor(s, m, WAVE);
    if (row < rows && l == 0 && !deferred) {
        const double r = alpha * s;
        y[row] = (beta == 0.0) ? r : fma(beta, y[row], r);
    }
    __syncthreads();
    for (int i = 0; i < n_long; ++i) { // (block-uniform; zero trips for all but a few blocks)
        const int lr = long_rows[i];
        const int p0 = rowptr[lr], p1 = rowptr[lr + 1];
        double t0 = 0.0, t1 = 0.0;
        int p = p0 + (int)threadIdx.x;
        for (; p + 256 < p1; p += 512) {
            const int c0 = colidx[p], c1 = colidx[p + 256];
            const double a0 = val[p], a1 = val[p + 256];
            t0 = fma(a0, x[c0], t0);
            t1 = fma(a1, x[c1], t1);
        }
        if (p < p1) t0 = fma(val[p], x[colidx[p]], t0);
        double t = t0 + t1;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) t += __shfl_xor(t, m, WAVE);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double r = alpha * (wsum[0] + wsum[1] + wsum[2] + wsum[3]);
            y[lr] = (beta == 0.0) ? r : fma(beta, y[lr], r);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// SpMV for short and medium rows (5..64 nonzeros per row on average), stream form.  The lanes-per-row kernel gives every row a lane group
// of 4..32 lanes: a 5-nonzero row keeps 4 of 8 lanes busy for two trips, and the stencil-like matrices that have such
// rows run at 2.6-3.3 TB/s.  Here a 256-thread block owns 256 consecutive rows, i.e. ONE contiguous run of nonzeros:
// all threads stream it (thread t takes nonzeros t, t + 256, ...; every lane busy, fully coalesced), park the
// products in LDS, and thread r then adds up the products of row r in CSR order.  A block whose rows hold more than
// the LDS can take (longer rows among the short ones) takes its rows in several runs.
// ---------------------------------------------------------------------------------------------
constexpr int ST_ROWS = 256;
// ST_CAP = products per block held in LDS: 6144 (48 KiB + skew: three blocks per CU) or 4096 (33 KiB: four).  A block walks
// its 256 rows in runs of up to ST_CAP products, and a block in its summing phase issues no loads, so what pays is the fewest
// runs first and the most blocks per CU second (round 3; 1 M banded rows of 7 / 13 / 27 per row: 29.5 / 43.6 / 97.1 us with
// 6144 against 24.2 / 36.6 / 82 us with 4096; 600 k rows of 48, three runs instead of two: 128 against 141 us): the
// launcher picks per call from the average row length (stream_cap).
constexpr int ST_LONG = 96;  // rows longer than this are summed by a whole wave
__device__ __forceinline__ int st_skew(int q) { return q + (q >> 5); } // rows of equal length: spread the LDS banks
template <int ST_CAP>
__global__ __launch_bounds__(ST_ROWS) void spmv_csr_stream_kernel(int rows, const int *__restrict__ rowptr,
                                                                 const int *__restrict__ colidx,
                                                                 const double *__restrict__ val,
                                                                 const double *__restrict__ x, double alpha, double beta,
                                                                 double *__restrict__ y)
{
    __shared__ double prod[ST_CAP + ST_CAP / 32 + 1];
    __shared__ int sp[ST_ROWS + 1];
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * ST_ROWS;
    const int nr = min(ST_ROWS, rows - row0);
    if (tid < nr) sp[tid] = rowptr[row0 + tid];
    if (tid == 0) sp[nr] = rowptr[row0 + nr];
    __syncthreads();
    __shared__ double wsum[ST_ROWS / 64];
    // The block's rows are taken in runs whose nonzeros fit the LDS: normally one run (all 256 rows); a block with
    // longer rows takes several, and a single row beyond the capacity is summed by the whole block.
    for (int r0 = 0; r0 < nr;) {
        const int base = sp[r0];
        int r1 = nr;
        if (sp[nr] - base > ST_CAP) { // largest r1 with sp[r1] - base <= ST_CAP (block-uniform: every thread searches)
            int lo = r0, hi = nr;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (sp[mid] - base <= ST_CAP) lo = mid; else hi = mid - 1;
            }
            r1 = lo;
        }
        if (r1 == r0) { // one row longer than the LDS capacity
            double sum = 0.0;
            for (int p = base + tid; p < sp[r0 + 1]; p += ST_ROWS) sum = fma(val[p], x[colidx[p]], sum);
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m, WAVE);
            if ((tid & 63) == 0) wsum[tid >> 6] = sum;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < ST_ROWS / 64; ++w) t += wsum[w];
                const double res = alpha * t;
                y[row0 + r0] = (beta == 0.0) ? res : fma(beta, y[row0 + r0], res);
            }
            __syncthreads();
            r0 += 1;
            continue;
        }
        const int total = sp[r1] - base;
        // UN nonzeros per thread in flight (clamped indices instead of predicates: no waits between the loads)
        constexpr int UN = 8;
        const int lastp = max(total - 1, 0);
        for (int p = tid; p < total; p += UN * ST_ROWS) {
            int c[UN];
            double a[UN], xv[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int q = min(p + u * ST_ROWS, lastp);
                c[u] = colidx[base + q];
                a[u] = val[base + q];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) xv[u] = x[c[u]];
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (p + u * ST_ROWS < total) prod[st_skew(p + u * ST_ROWS)] = a[u] * xv[u];
        }
        __syncthreads();
        // thread r adds up row r in CSR order -- unless the row is long (skewed matrices: one thread walking a
        // 5000-entry row holds the whole block for ~150 us): those are summed by the 64 lanes of the row's wave
        const bool mine = tid >= r0 && tid < r1;
        const int q0 = mine ? sp[tid] - base : 0, q1 = mine ? sp[tid + 1] - base : 0;
        const bool longrow = q1 - q0 > ST_LONG;
        if (mine && !longrow) {
            double sum = 0.0;
            for (int q = q0; q < q1; ++q) sum += prod[st_skew(q)];
            const double res = alpha * sum;
            y[row0 + tid] = (beta == 0.0) ? res : fma(beta, y[row0 + tid], res);
        }
        for (unsigned long long lm = __builtin_amdgcn_ballot_w64(longrow); lm; lm &= lm - 1) {
            const int src = __builtin_ctzll(lm);                 // lane whose row this is
            const int a = __shfl(q0, src, WAVE), b = __shfl(q1, src, WAVE);
            double sum = 0.0;
            for (int q = a + (tid & 63); q < b; q += WAVE) sum += prod[st_skew(q)];
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m, WAVE);
            if ((tid & 63) == src) {
                const double res = alpha * sum;
                y[row0 + tid] = (beta == 0.0) ? res : fma(beta, y[row0 + tid], res);
            }
        }
        if (r1 < nr) __syncthreads(); // prod is reused by the next run
        r0 = r1;
    }
}

// ---------------------------------------------------------------------------------------------
// SpMV for medium rows (49..96 nonzeros), segmented form.  With one row per wave a 73-nonzero row (Queen_4147) fills
// 57 % of two 64-lane slices and walks two dependent trips: 2.3 TB/s.  Here a wave owns R consecutive rows -- one
// contiguous run of nonzeros -- and streams it in unpredicated slices of 64 (clamped indices, S slices in flight:
// 91 % of the lanes busy for R = 4, S = 5 at 73 per row); every lane knows the row of its entry from the R + 1 row
// pointers (wave-uniform after a readlane), products are accumulated per row and folded once at the end: the
// wave-level segmented reduction of the north star.
// ---------------------------------------------------------------------------------------------
template <int R, int S>
__global__ __launch_bounds__(256) void spmv_csr_seg_kernel(int rows, const int *__restrict__ rowptr,
                                                          const int *__restrict__ colidx,
                                                          const double *__restrict__ val,
                                                          const double *__restrict__ x, double alpha, double beta,
                                                          double *__restrict__ y)
{
    static_assert(R >= 1 && R <= 16, "row pointers are broadcast from the first R + 1 lanes");
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (r0 >= rows) return;
    const int mine = rowptr[min(r0 + min(lane, R), rows)];
    int b[R + 1];
#pragma unroll
    for (int i = 0; i <= R; ++i) b[i] = __builtin_amdgcn_readlane(mine, i);
    const int p0 = b[0], p1 = b[R], last = p1 - 1;
    double acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.0;
    for (int base = p0; base < p1; base += S * WAVE) {
        int c[S];
        double a[S], xv[S];
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int p = min(base + u * WAVE + lane, last);
            c[u] = colidx[p];
            a[u] = val[p];
        }
#pragma unroll
        for (int u = 0; u < S; ++u) xv[u] = x[c[u]];
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int idx = base + u * WAVE + lane;
            const double prod = (idx <= last) ? a[u] * xv[u] : 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) acc[i] += (idx >= b[i] && idx < b[i + 1]) ? prod : 0.0;
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        double sum = acc[i];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m, WAVE);
        if (lane == i && r0 + i < rows) {
            const double res = alpha * sum;
            y[r0 + i] = (beta == 0.0) ? res : fma(beta, y[r0 + i], res);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SpMV for long rows, x window in LDS (second attempt).  Diagnostics on the plain kernel: the A stream alone runs at
// 6.8 TB/s with the same row-per-wave shape (tools/stream_bench.hip), replacing the gather by a one-line read still
// leaves 76 us -- what costs is the second, dependent vector-memory access per slice (address unit ~15 cycles per
// instruction, more for a 40-line gather).  Here an 8-row block (one row per wave; 16 rows in round 2) fetches the x range its rows span
// into LDS once and gathers from there; the stream loads of a row (up to 448 nonzeros) are issued right after its row
// pointers, BEFORE the window is known, so the block-wide min/max, the window load and their three barriers hide
// behind the HBM latency of the stream (the barriers are `s_barrier` without the vmcnt(0) of __syncthreads).
// Columns outside the window (unsorted rows) are fetched from global memory lane by lane.
// ---------------------------------------------------------------------------------------------
// Eight waves per block and 36 KiB of window: FOUR blocks per CU, each in another phase of its life (row pointers, stream
// in flight, window fetch, gathers, reduction), so that the CU's HBM requests do not come in two bursts.  Sixteen waves and
// 40 KiB (two blocks per CU, round 2): bench matrix 68.3 us against 63.5-64.1 us now; 600 k banded rows of 160 / 260: 345 /
// 489 us against 337 / 470; 300 k rows of 500 over +-3000 (span beyond the window either way): 529 against 489.
constexpr int SPMV_LDS_ROWS = 8;    // = waves per block
constexpr int SPMV_LDS_THREADS = SPMV_LDS_ROWS * 64;
constexpr int SPMV_LDS_CAP = 4608;  // doubles (36 KiB): four blocks per CU
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int RW, int S> // RW rows per wave (16 RW rows per block), S slices of 64 nonzeros fetched ahead per row
__global__ __launch_bounds__(SPMV_LDS_THREADS) void spmv_csr_lds_kernel(int rows, int cols, const int *__restrict__ rowptr,
                                                           const int *__restrict__ colidx,
                                                           const double *__restrict__ val,
                                                           const double *__restrict__ x, double alpha, double beta,
                                                           double *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) double xs[];
    __shared__ int sm_lo, sm_hi;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    const int row0 = (blockIdx.x * SPMV_LDS_ROWS + wave) * RW;
    if (tid == 0) {
        sm_lo = 0x7fffffff;
        sm_hi = -1;
    }
    int p0[RW], last[RW];
    bool has[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        int a0 = 0, a1 = 0;
        if (row0 + r < rows) {
            a0 = wave_uniform(rowptr[row0 + r]);
            a1 = wave_uniform(rowptr[row0 + r + 1]);
        }
        p0[r] = a0;
        has[r] = a1 > a0;
        last[r] = max(a1 - 1, a0);
    }
    // the two ends of every row first (lane 2r: first column of row r, lane 2r+1: its last column), then the first
    // burst of the streams
    int c[RW][S];
    double a[RW][S];
    int ce = 0;
    bool ce_valid = false;
#pragma unroll
    for (int r = 0; r < RW; ++r)
        if (has[r] && (lane >> 1) == r) {
            ce = colidx[(lane & 1) ? last[r] : p0[r]];
            ce_valid = true;
        }
#pragma unroll
    for (int r = 0; r < RW; ++r)
        if (has[r]) {
#pragma unroll
            for (int u = 0; u < S; ++u) {
                const int p = min(p0[r] + u * WAVE + lane, last[r]);
                c[r][u] = colidx[p];
                a[r][u] = val[p];
            }
        }
    lds_barrier(); // sm_lo / sm_hi initialised
    if (ce_valid) {
        if (lane & 1) atomicMax(&sm_hi, ce);
        else atomicMin(&sm_lo, ce);
    }
    lds_barrier();
    int lo = sm_lo, hi = sm_hi;
    if (lo > hi) {
        lo = 0;
        hi = -1;
    }
    lo = max(lo, 0);
    hi = min(hi, cols - 1);
    // a span that does not fit is not staged at all: every gather then goes to global memory, as in the plain kernel.
    // The window starts at an even column so that it can be fetched by LDS-DMA in 16-byte pieces (x 16-byte aligned,
    // the last pair inside x); otherwise eight bytes per thread through registers.
    lo &= ~1;
    const int wlen = (hi - lo + 1 <= SPMV_LDS_CAP) ? hi - lo + 1 : 0;
    const int pairs = (wlen + 1) >> 1;
    if (wlen > 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && lo + 2 * pairs <= cols) {
        const char *src = reinterpret_cast<const char *>(x + lo);
        for (int p0 = wave * 64; p0 < pairs; p0 += SPMV_LDS_THREADS) { // (wave-uniform trip count)
            const int pr = min(p0 + lane, pairs - 1);  // clamped lanes rewrite the last pair into the slack area
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)pr * 16),
                                             (__attribute__((address_space(3))) void *)(xs + 2 * p0), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (wlen > 0) {
        // the whole window in one burst of loads (clamped indices), then the stores
        constexpr int PASSES = SPMV_LDS_CAP / SPMV_LDS_THREADS;
        double t[PASSES];
#pragma unroll
        for (int j = 0; j < PASSES; ++j) t[j] = x[lo + min(tid + SPMV_LDS_THREADS * j, wlen - 1)];
#pragma unroll
        for (int j = 0; j < PASSES; ++j)
            if (tid + SPMV_LDS_THREADS * j < wlen) xs[tid + SPMV_LDS_THREADS * j] = t[j];
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = row0 + r;
        if (row >= rows) break;
        double s0 = 0.0, s1 = 0.0;
        if (has[r]) {
            for (int base = p0[r];;) {
                double xv[S];
                bool out = false;
#pragma unroll
                for (int u = 0; u < S; ++u) {
                    const bool live = base + u * WAVE + lane <= last[r];
                    const unsigned rel = (unsigned)(c[r][u] - lo);
                    const bool inw = rel < (unsigned)wlen;
                    // never multiply by unstaged LDS (a block whose span is not staged has wlen = 0) nor, in a clamped
                    // lane, by a staged Inf / NaN that belongs to another entry: dead lanes contribute exactly 0 * 0
                    const double staged = xs[inw ? rel : 0u];
                    xv[u] = (inw && live) ? staged : 0.0;
                    out |= live && !inw;
                }
                if (__builtin_amdgcn_ballot_w64(out) != 0ull) { // columns outside the window (unsorted rows, wide spans)
#pragma unroll
                    for (int u = 0; u < S; ++u) {
                        const bool live = base + u * WAVE + lane <= last[r];
                        if (live && (unsigned)(c[r][u] - lo) >= (unsigned)wlen) xv[u] = x[c[r][u]];
                    }
                }
#pragma unroll
                for (int u = 0; u < S; ++u) {
                    const double av = (base + u * WAVE + lane <= last[r]) ? a[r][u] : 0.0;
                    if (u & 1) s1 = fma(av, xv[u], s1);
                    else s0 = fma(av, xv[u], s0);
                }
                base += S * WAVE;
                if (base > last[r]) break;
#pragma unroll
                for (int u = 0; u < S; ++u) {
                    const int p = min(base + u * WAVE + lane, last[r]);
                    c[r][u] = colidx[p];
                    a[r][u] = val[p];
                }
            }
        }
        double sum = s0 + s1;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m, WAVE);
        if (lane == 0) {
            const double res = alpha * sum;
            y[row] = (beta == 0.0) ? res : fma(beta, y[row], res);
        }
    }
}


// the stream kernel with the LDS capacity that gives a block of average rows the fewest runs; a tie goes to the smaller one
static hipError_t launch_stream(hipStream_t s, int rows, double avg, const int *rowptr, const int *colidx, const double *val,
                                const double *x, double alpha, double beta, double *y)
{
    const double per_block = avg * ST_ROWS;
    const int runs4 = (int)((per_block + 4095.0) / 4096.0), runs6 = (int)((per_block + 6143.0) / 6144.0);
    const dim3 grid((unsigned)((rows + ST_ROWS - 1) / ST_ROWS));
    if (runs4 <= runs6)
        hipLaunchKernelGGL(spmv_csr_stream_kernel<4096>, grid, dim3(ST_ROWS), 0, s, rows, rowptr, colidx, val, x, alpha, beta, y);
    else
        hipLaunchKernelGGL(spmv_csr_stream_kernel<6144>, grid, dim3(ST_ROWS), 0, s, rows, rowptr, colidx, val, x, alpha, beta, y);
    return hipGetLastError();
}

template <int LPR>
static hipError_t spmv_go(hipStream_t s, int rows, const int *rowptr, const int *colidx, const double *val,
                          const double *x, double alpha, double beta, double *y)
{
    constexpr int rpb = 256 / LPR;
    hipLaunchKernelGGL(spmv_csr_kernel<LPR>, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, s, rows,
                       rowptr, colidx, val, x, alpha, beta, y);
    return hipGetLastError();
}

hipError_t launch_spmv(hipStream_t s, int rows, int cols, int64_t nnz, const int *rowptr, const int *colidx,
                       const double *val, const double *x, double alpha, double beta, double *y)
{
    const double avg = rows > 0 ? (double)nnz / (double)rows : 0.0;
    const char *sv = options().spmv_variant; // "" = auto; anything else pins a kernel (A/B runs, tests)
    const bool autosel = !*sv;
    auto is = [&](const char *name) { return !strcmp(sv, name); };
#define SBLAS_SPMV_LDS(RWV, SV)                                                                                      \
    do {                                                                                                             \
        raise_dynamic_lds((const void *)spmv_csr_lds_kernel<RWV, SV>, (SPMV_LDS_CAP + 128) * sizeof(double));        \
        hipLaunchKernelGGL((spmv_csr_lds_kernel<RWV, SV>),                                                           \
                           dim3((unsigned)((rows + SPMV_LDS_ROWS * RWV - 1) / (SPMV_LDS_ROWS * RWV))), dim3(SPMV_LDS_THREADS),    \
                           (SPMV_LDS_CAP + 128) * sizeof(double), s, rows, cols, rowptr, colidx, val, x, alpha, beta, y); \
        return hipGetLastError();                                                                                    \
    } while (0)
#define SBLAS_SPMV_SEG(RV, SV)                                                                                       \
    do {                                                                                                             \
        hipLaunchKernelGGL((spmv_csr_seg_kernel<RV, SV>), dim3((unsigned)((rows + 4 * RV - 1) / (4 * RV))), dim3(256), \
                           0, s, rows, rowptr, colidx, val, x, alpha, beta, y);                                      \
        return hipGetLastError();                                                                                    \
    } while (0)
    if (autosel) {
        // long rows: x window in LDS (bench matrix: 70-73 us vs 82-85 us for the lanes-per-row kernel); a block whose
        // rows span more than the LDS window degrades to global gathers by itself.  Slices in flight per row: ~1.3-1.5 x
        // the row length in 64-lane slices (600 k banded rows of 100 / 130 / 160 / 200 / 260, band +-2000: S = 2 / 3 / 3 /
        // 4 / 7 take 273 / 307 / 315 / 360 / 433 us against 329 / 337 / 345 / 360 / 451 us with S = 4 throughout; the
        // same order on a +-20000 band, tools/spmv_rowlen_sweep.py)
        if (avg > 96.0) {
            if (avg <= 115.0) SBLAS_SPMV_LDS(1, 2);
            if (avg <= 180.0) SBLAS_SPMV_LDS(1, 3);
            if (avg <= 230.0) SBLAS_SPMV_LDS(1, 4);
            SBLAS_SPMV_LDS(1, 7);
        }
        // medium rows: R rows per wave, segmented (Queen-like rows, 73 per row: 232 us vs 395 us; banded synthetic rows
        // of 36 / 72 / 90: 122 / 266 / 351 us vs 150 / 339 / 375 us for the lanes-per-row kernel)
        if (avg > 64.0) SBLAS_SPMV_SEG(4, 5);
        // short and medium rows (5 < avg <= 64): 256 rows per block streamed through LDS, in runs of up to 6144
        // products (stencil-like rows of 7 / 13 / 27: 108 / 177 / 344 us vs 143 / 277 / 498 us for the lanes-per-row and
        // segmented kernels; banded-random rows of 14 / 20 / 28 / 36 / 48: 46 / 62 / 85 / 116 / 161 vs 49 / 71 / 94 /
        // 128 / 194; 1 M banded rows of 55 / 70: 207 / 256 us vs 250 / 281 us segmented; Queen-like rows of 73: 251 vs
        // 256 us).  Rows beyond 96 take the kernel's slow path (a wave per row), so it stops where a spread of row
        // lengths starts to reach that: Poisson rows of 60 on average tie, of 70 lose 4 %, of 80 7 %, of 90 27 % -- the launcher
        // only knows the average.  Round 3 (four blocks per CU for short rows): 2 M uniform rows of 3 / 5: 35.9 / 40.0 us
        // against 34.6-36.1 / 50.0 us for the lanes-per-row kernel, power-law rows averaging 3.2: 55.7 against 67 us -- the
        // stream form from 2.5 per row on (round 2: from 5).
        if (avg > 2.5) {
            return launch_stream(s, rows, avg, rowptr, colidx, val, x, alpha, beta, y);
        }
        return spmv_go<4>(s, rows, rowptr, colidx, val, x, alpha, beta, y);
    }
    if (is("lds")) SBLAS_SPMV_LDS(1, 7);
    if (is("lds2")) SBLAS_SPMV_LDS(2, 7);
    if (is("lds1s2")) SBLAS_SPMV_LDS(1, 2);
    if (is("lds1s3")) SBLAS_SPMV_LDS(1, 3);
    if (is("lds1s4")) SBLAS_SPMV_LDS(1, 4);
    if (is("seg4")) SBLAS_SPMV_SEG(4, 5);
    if (is("seg3")) SBLAS_SPMV_SEG(3, 4);
    if (is("seg8")) SBLAS_SPMV_SEG(8, 5);
    if (is("seg2")) SBLAS_SPMV_SEG(2, 3);
#undef SBLAS_SPMV_LDS
#undef SBLAS_SPMV_SEG
    if (is("stream")) {
        return launch_stream(s, rows, avg, rowptr, colidx, val, x, alpha, beta, y);
    }
    // "plain" (and anything unknown): lanes per row by average length
    if (avg <= 6.0) return spmv_go<4>(s, rows, rowptr, colidx, val, x, alpha, beta, y);
    if (avg <= 12.0) return spmv_go<8>(s, rows, rowptr, colidx, val, x, alpha, beta, y);
    if (avg <= 24.0) return spmv_go<16>(s, rows, rowptr, colidx, val, x, alpha, beta, y);
    if (avg <= 48.0) return spmv_go<32>(s, rows, rowptr, colidx, val, x, alpha, beta, y);
    return spmv_go<64>(s, rows, rowptr, colidx, val, x, alpha, beta, y);
}

} // namespace sblas
