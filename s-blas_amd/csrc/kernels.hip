// kernels.hip -- hand-written gfx950 (CDNA4, wave64) SpMM kernels of the S-BLAS CSR hot path.
//
// SpMM  C = alpha*A*B + beta*C   (stage 1 + stage 2; the launcher at the end of this file picks the kernels)
//   dense_to_rowmajor_kernel     B (col-major) -> Bt (row-major, zero padded, one all-zero row)          [stage 1]
//   stage_classify_kernel        stage 1 + classify_panels_kernel in one launch (fused C-ABI entry)
//   classify_panels_kernel       per row panel: column span and class (LDS-tiled / direct / matrix cores), shared-rows flag
//   mfma_vote_kernel             one workgroup: matrix-wide decisions before stage 2 (128+ staged columns)
//   spmm_window6_kernel<G, NH>   qualifying panels: 152-row x 64-column B tiles through LDS (LDS-DMA loader waves),
//                                one DPP row per matrix row, streaming windows of A
//   spmm_direct_dpp_kernel<GROUPS> all other panels: a row per wave, Bt rows straight from L2, DPP broadcast
//   spmm_direct_merge_kernel     ... of matrices whose neighbouring rows share column patterns (multi-dof FEM, 128-column
//                                tiles): three rows per wave, shared Bt loads
//   (spmm_mfma_kernel, panels of dense 16 x 4 sub-blocks on the fp64 matrix cores: spmm_mfma.hip)
//   spmm_direct_rows_kernel      direct panels of short-row matrices (< 56 per row at 64 columns), and from 128 columns on
//                                wherever the classifier's vote prefers it: four rows per wave
//   spmm_rowpanel_narrow_kernel  n <= 8 (sub-wave lane groups; 16 / 32 columns behind SBLAS_SPMM_MIN_LDBT=0)
//   spmm_rows8_kernel            n <= 8 and rows of 256+ nonzeros on average: a wave per row, eight sums per lane
// Epilogues and merges
//   axpby_kernel                 y = beta*y + alpha*x                                        (kernel.h:27-38)
//   scale_kernel                 C = beta*C (a matrix without nonzeros)
//   merge_rowblocks_kernel       method-2 / SpMV merge: scatter packed row blocks, apply alpha / beta
// (SpMV kernels: spmv_kernels.hip.)
//
// These replace the closed-source cuSPARSE calls of the reference (spmm.h:146-149, :248-251) and its one utility
// kernel (kernel.h:27-38).  Everything is written for 64-wide wavefronts; there is no 32-lane code path.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <string.h>
#include <stdio.h>
#include <mutex>
#include <vector>
#include "kernels.h"

namespace sblas {

constexpr int WAVE = 64;

__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ double readlane_f64(double v, int src_lane)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// Stage 1: Bt[k][j] = B[k + j*ldb]   (k < cols, j < n), Bt[k][j] = 0 for n <= j < ldbt.
// 32 (k) x 64 (j) tile through LDS: global reads run along k (contiguous in col-major B, 256 bytes per half-wave),
// global writes run along j (contiguous in row-major Bt, 512 bytes per wave).  Row stride 33 doubles keeps both LDS
// phases conflict-free for ds_read/write_b64.  17 KB of LDS per workgroup: nine workgroups per CU, so the 2252 tiles
// of the bench shape are resident at once (64 x 64 tiles, 33 KB: four per CU, 1126 tiles = one round and a tenth).
// ---------------------------------------------------------------------------------------------
constexpr int STAGE_K = 32;
// While it is at it the pass looks for non-finite values in B: tail[TAIL_NONFINITE] = stage_epoch when it meets one
// (every writer stores the same value), and tail[TAIL_STAGE_EPOCH] = stage_epoch always -- the two agree exactly when
// the staging copy the stage-2 kernels are about to read holds an Inf or NaN (the MFMA kernel then leaves its panels
// to the vector kernels: 0 * Inf from a block's zero fill would reach rows that never refer to that row of B).
__device__ __forceinline__ void stage_tile(double (*tile)[STAGE_K + 1], int64_t k0, int64_t j0, int64_t cols, int64_t n,
                                           const double *__restrict__ B, int64_t ldb, double *__restrict__ Bt,
                                           int64_t ldbt, int *__restrict__ tail, int stage_epoch)
{
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    // all eight loads of a thread in flight before the first LDS store (the kernel is latency-bound otherwise)
    double v[8];
    const int64_t k = k0 + tx;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int64_t j = j0 + ty + 8 * u;
        v[u] = (j < n && k < cols) ? B[k + j * ldb] : 0.0;
    }
    bool odd = false;
#pragma unroll
    for (int u = 0; u < 8; ++u) odd |= (__double2hiint(v[u]) & 0x7ff00000) == 0x7ff00000;
    if (__builtin_amdgcn_ballot_w64(odd) != 0ull && (threadIdx.x & 63) == 0) tail[TAIL_NONFINITE] = stage_epoch;
    if (k0 == 0 && j0 == 0 && threadIdx.x == 0) tail[TAIL_STAGE_EPOCH] = stage_epoch;
#pragma unroll
    for (int u = 0; u < 8; ++u) tile[ty + 8 * u][tx] = v[u];
    __syncthreads();
    const int jl = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int64_t j = j0 + jl;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int kk = kq + 4 * u;
        const int64_t kr = k0 + kk;
        if (kr < cols && j < ldbt) Bt[kr * ldbt + j] = tile[jl][kk];
        if (kr == cols && j < ldbt) Bt[kr * ldbt + j] = 0.0; // the all-zero row masked DPP slots point at
    }
}
__global__ __launch_bounds__(256) void dense_to_rowmajor_kernel(int64_t cols, int64_t n,
                                                               const double *__restrict__ B, int64_t ldb,
                                                               double *__restrict__ Bt, int64_t ldbt,
                                                               int *__restrict__ tail, int stage_epoch)
{
    __shared__ double tile[64][STAGE_K + 1];
    stage_tile(tile, (int64_t)blockIdx.x * STAGE_K, (int64_t)blockIdx.y * 64, cols, n, B, ldb, Bt, ldbt, tail,
               stage_epoch);
}

// Stage 1 for narrow blocks (ldbt = NC in {8, 16, 32}): a thread per row of Bt.  The reads of a wave run along k
// (512 contiguous bytes per column of B), every thread writes its NC * 8 contiguous bytes of Bt, a wave 64 such rows
// in a row.  (The 32 x 64 tile transposer above spends the time of a 64-column block on any narrower one: 16 us on the
// bench shape against 2-3 us here.)
template <int NC>
__device__ __forceinline__ void stage_rows_narrow(int64_t k0, int64_t cols, int64_t n, const double *__restrict__ B,
                                                  int64_t ldb, double *__restrict__ Bt, int *__restrict__ tail,
                                                  int stage_epoch)
{
    const int64_t k = k0 + threadIdx.x;
    double v[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) v[j] = (j < n && k < cols) ? B[k + (int64_t)j * ldb] : 0.0;
    bool odd = false;
#pragma unroll
    for (int j = 0; j < NC; ++j) odd |= (__double2hiint(v[j]) & 0x7ff00000) == 0x7ff00000;
    if (__builtin_amdgcn_ballot_w64(odd) != 0ull && (threadIdx.x & 63) == 0) tail[TAIL_NONFINITE] = stage_epoch;
    if (k0 == 0 && threadIdx.x == 0) tail[TAIL_STAGE_EPOCH] = stage_epoch;
    if (k <= cols) { // row `cols` is the all-zero row
        double2 *dst = reinterpret_cast<double2 *>(Bt + k * NC);
#pragma unroll
        for (int j = 0; j < NC / 2; ++j) dst[j] = make_double2(v[2 * j], v[2 * j + 1]);
    }
}
template <int NC>
__global__ __launch_bounds__(256) void dense_to_rowmajor_narrow_kernel(int64_t cols, int64_t n,
                                                                      const double *__restrict__ B, int64_t ldb,
                                                                      double *__restrict__ Bt,
                                                                      int *__restrict__ tail, int stage_epoch)
{
    stage_rows_narrow<NC>((int64_t)blockIdx.x * 256, cols, n, B, ldb, Bt, tail, stage_epoch);
}

// Stage 1 of a ROW BLOCK (method 2: rows << cols).  A block of a banded matrix refers to a narrow range of columns,
// so only that range of B needs a row-major copy: at config 5's shape a rank of eight stages 15 % of B instead of
// all of it.  colrange_kernel finds the exact range (every column index is read: rows need not be sorted), one
// (min, max) per workgroup into the span array the classifier fills later; stage_range_kernel folds those and walks
// the 32-row tiles of the range, plus the tile holding the all-zero row, in a grid-stride loop.  Rows of Bt outside
// the range keep whatever the workspace held: no nonzero points at them, and no kernel multiplies a row of Bt that
// no nonzero points at (masked slots go to the zero row).
__device__ __forceinline__ void colrange_part(int64_t nnz, const int *__restrict__ colidx, int2 *__restrict__ part,
                                              int block, int nblocks, int2 *red)
{
    int lo = INT_MAX, hi = -1;
    const int64_t stride = (int64_t)nblocks * 1024;
    for (int64_t i0 = (int64_t)block * 1024 + threadIdx.x; i0 < nnz; i0 += stride) {
        int c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = (i0 + 256 * u < nnz) ? colidx[i0 + 256 * u] : INT_MAX;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lo = min(lo, c[u]);
            hi = max(hi, c[u] == INT_MAX ? -1 : c[u]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o));
        hi = max(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_int2(lo, hi);
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            lo = min(lo, red[w].x);
            hi = max(hi, red[w].y);
        }
        part[block] = make_int2(lo, hi);
    }
}
__global__ __launch_bounds__(256) void colrange_kernel(int64_t nnz, const int *__restrict__ colidx,
                                                       int2 *__restrict__ part)
{
    __shared__ int2 red[4];
    colrange_part(nnz, colidx, part, (int)blockIdx.x, (int)gridDim.x, red);
}
__global__ __launch_bounds__(256) void stage_range_kernel(int64_t cols, int64_t n, const double *__restrict__ B,
                                                          int64_t ldb, double *__restrict__ Bt, int64_t ldbt,
                                                          int *__restrict__ tail, const int2 *__restrict__ part,
                                                          int nparts, int stage_epoch)
{
    __shared__ double tile[64][STAGE_K + 1];
    __shared__ int2 red[4];
    int lo = INT_MAX, hi = -1;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        const int2 p = part[i];
        lo = min(lo, p.x);
        hi = max(hi, p.y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o));
        hi = max(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_int2(lo, hi);
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        lo = min(lo, red[w].x);
        hi = max(hi, red[w].y);
    }
    lo = max(lo, 0); // indices outside [0, cols) are the caller's error; they must not become wild tile numbers
    hi = (int)min((int64_t)hi, cols - 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) tail[TAIL_STAGE_EPOCH] = stage_epoch;
    const int kz = (int)(cols / STAGE_K); // the tile of the all-zero row (row index cols)
    int kt0 = lo / STAGE_K, kt1 = hi / STAGE_K;
    if (hi < lo) kt0 = kz, kt1 = kz;
    const int extra = (kz > kt1 || kz < kt0) ? 1 : 0;
    const int ny = (int)((ldbt + 63) / 64);
    const int64_t total = (int64_t)(kt1 - kt0 + 1 + extra) * ny;
    for (int64_t t = blockIdx.x; t < total; t += gridDim.x) {
        const int ky = (int)(t % ny);
        int kt = kt0 + (int)(t / ny);
        if (kt > kt1) kt = kz;
        // stage_tile's own epoch store fires for tile (0, 0) only; it writes the same value
        stage_tile(tile, (int64_t)kt * STAGE_K, (int64_t)ky * 64, cols, n, B, ldb, Bt, ldbt, tail, stage_epoch);
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------
// Stage 2, wide form (ldbt a multiple of 64).
// Workgroup = 4 waves = one panel of PANEL_ROWS consecutive rows x one 64-column tile of C.
// A wave owns whole rows; its 64 lanes are the 64 columns of the tile, so
//   - row_ptr / col_idx / val of a row are read once per wave, 64 nonzeros per coalesced load
//     (lane l takes nonzero p+l), and handed to all lanes through v_readlane (the column index and
//     the value become scalars -> the B address is scalar base + lane*8, one full 512-byte row of
//     Bt per nonzero, perfectly coalesced);
//   - the accumulator is one fp64 register per lane, summed in CSR order (same order as the
//     reference's CPU loop, spmm.h:59-64);
//   - finished rows are parked in an LDS tile [column][row] and the panel is written back with
//     lanes running along the row index, which is the contiguous direction of column-major C
//     (this is also where alpha/beta are applied, so C is read and written exactly once).
// ---------------------------------------------------------------------------------------------
constexpr int PANEL_ROWS = 32;   // narrow kernels (256 threads)

// Wide kernel geometry: 16 waves (1024 threads), one row per wave at a time -> a 16-row panel.
// Few rows in flight per CU keeps the set of Bt rows that the resident workgroups of one XCD touch
// (rows in flight + the matrix band) inside that XCD's 4 MiB L2; with 32-row panels on 256-thread
// blocks ~65 000 rows were in flight chip-wide and every Bt row came from the Infinity Cache.
constexpr int WIDE_WAVES = 16;
constexpr int WIDE_ROWS_PER_WAVE = 1;
constexpr int WIDE_PANEL = WIDE_WAVES * WIDE_ROWS_PER_WAVE;
constexpr int DPP_LONG = 4096;            // entries from which a row of the row-per-wave kernel is computed by the whole workgroup

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one).  Give every XCD one contiguous
// range of panels so that neighbouring panels -- which read overlapping Bt rows -- share an L2 (speed only;
// any placement is correct).  Bijective for every panel count.
__device__ __forceinline__ int xcd_contiguous_panel(int b, int npanels)
{
    const int xcd = b & 7, idx = b >> 3;
    const int q = npanels >> 3, r = npanels & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}


// panel census of the windowed kernel: [0] windowed, [1] direct (too sparse over its span), [2] windowed but
// recomputed by the fallback.  One atomic per panel; read through sblas_hip_debug_spmm_panel_stats.
__device__ unsigned long long g_panel_stats[4];

// one row, straight from Bt in L2 (also the per-panel fallback of the windowed kernel)
__device__ __forceinline__ double row_direct(const int *__restrict__ colidx, const double *__restrict__ val,
                                             const double *__restrict__ Bt, unsigned ld32, unsigned lane_off,
                                             int lane, int p0, int p1)
{
    double acc = 0.0;
    for (int p = p0; p < p1; p += WAVE) {
        const int mine = p + lane;
        int cj = 0;
        double vj = 0.0;
        if (mine < p1) {
            cj = colidx[mine];
            vj = val[mine];
        }
        const int cnt = min(WAVE, p1 - p);
        int k = 0;
        for (; k + 8 <= cnt; k += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = __builtin_amdgcn_readlane(cj, k + u);
                const double a = readlane_f64(vj, k + u);
                acc = fma(a, (Bt + (size_t)((unsigned)c * ld32))[lane_off], acc);
            }
        }
        for (; k < cnt; ++k) {
            const int c = __builtin_amdgcn_readlane(cj, k);
            const double a = readlane_f64(vj, k);
            acc = fma(a, (Bt + (size_t)((unsigned)c * ld32))[lane_off], acc);
        }
    }
    return acc;
}
#ifndef SBLAS_W2_ROWS
#define SBLAS_W2_ROWS 152
#endif
// 152 rows: two tiles fill the 160 KiB of LDS (128-row tiles: 0.266 ms per step on the bench matrix, 144: 0.252, 152: 0.248 --
// fewer (rows, tile) visits, fuller 4-step blocks; gpurun_out/r3_ab_tile.txt)
constexpr int W2_ROWS = SBLAS_W2_ROWS;    // Bt rows per LDS tile (a multiple of 8: four loader waves, two rows per DMA instruction)
constexpr int W2_LROWS = W2_ROWS / 4;     // ... per loader wave
static_assert(W2_ROWS % 8 == 0, "tile rows");
constexpr int W2_TILE = W2_ROWS * 64;     // doubles
constexpr size_t W2_LDS_BYTES = (2 * (size_t)W2_TILE + 64) * sizeof(double) + 64 * sizeof(int);

// The panel classifier.  One wave per panel, one lane per row.  Verdicts (workspace tail, kernels.h):
//   info[p] = column span (first, last) of panel p, (1, 0) for an empty panel;
//   cls[p]  = PANEL_WINDOW  dense enough over its span for the LDS-tiled kernel,
//             PANEL_MFMA_W / PANEL_MFMA_D  its nonzeros sit in dense 16 x 4 sub-blocks (sampled: fill of the blocks
//             its first 16 rows touch >= mfma_min_fill): the matrix-core kernel, and the windowed (W) or direct (D)
//             kernel should stage 1 have met a non-finite B,
//             PANEL_DIRECT  everything else.
// bitmap: MFMA_BITMAP_WORDS ints of LDS per wave (one bit per 4-column block of the sampled rows' span).
constexpr int MFMA_BITMAP_WORDS = 1024; // 32768 blocks = 131072 columns of span
__device__ __forceinline__ void classify_panel(int p, int rows, int cols, int npanels, int panel_rows,
                                               const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                               int max_row_len, float min_density, float min_rowlen, float mfma_min_fill,
                                               int merge_probe, int *__restrict__ tail, int2 *__restrict__ info,
                                               int *__restrict__ cls, int epoch, unsigned *__restrict__ bitmap)
{
    const int lane = threadIdx.x & 63;
    if (p >= npanels) return;
    int first = 0x7fffffff, last = -1, len = 0, mlen = 0;
    for (int rr = lane; rr < panel_rows; rr += WAVE) { // panels of up to 144 rows: up to three rows per lane
        const int row = p * panel_rows + rr;
        if (row < rows) {
            const int a = rowptr[row], b = rowptr[row + 1];
            len += b - a;
            mlen = max(mlen, b - a);
            if (b > a) {
                first = min(first, colidx[a]);
                last = max(last, colidx[b - 1]);
            }
        }
    }
    int nnz = len;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        first = min(first, __shfl_xor(first, m, WAVE));
        last = max(last, __shfl_xor(last, m, WAVE));
        nnz += __shfl_xor(nnz, m, WAVE);
        mlen = max(mlen, __shfl_xor(mlen, m, WAVE));
    }
    const bool sane = last >= first && first >= 0 && last < cols; // (first / last of sorted rows; the kernels re-check)
    // (narrow widths: a visit of the lane-per-entry kernel costs the same whether a row has 5 or 30 entries in the tile, so
    //  short rows stay with the direct kernels: min_rowlen = nonzeros per row the panel must average)
    bool window_ok = sane && mlen <= max_row_len && (float)nnz >= min_density * (float)(last - first + 1) &&
                     (float)nnz >= min_rowlen * (float)min(panel_rows, rows - p * panel_rows);
    // Do the rows come in groups that list the same columns (the unknowns of one mesh node in a multi-dof FEM matrix),
    // and is the group length a multiple of three (3-dof nodes; six dofs are two groups of three)?  Neighbouring rows
    // at the panel's head are compared entry by entry: the first pair that differs ends the (possibly cut) group the
    // panel starts in, the pairs that agree behind it measure the next group.  Only lengths 3, 6, 9 feed the
    // row-merging kernel three rows at a time throughout (groups of four merge half of their triples and lose to the
    // other kernels; pairs do not merge at all); (row index of a group start) mod 3 rides in the class word, so that
    // the merging kernel lines its waves up with the groups -- a method-2 row block begins anywhere.
    // Probed where the row-merging kernel is a choice (128+ staged columns): on the panels the LDS-tiled kernel cannot
    // take, and on those it would take only under the lowered bar -- grid-structured rows of three unknowns per node,
    // band +-1000, N = 256: 1.32 ms merged against 2.32 ms through LDS -- which then go to the direct path.
    bool shared = false;
    int phase = 0;
    const bool strong = window_ok && (float)nnz >= (float)panel_rows / 16.0f * (float)(last - first + 1);
    if (sane && !strong && (merge_probe & 1)) {
        const int r0 = p * panel_rows;
        int start = -1, len = 0; // first row of the measured group (relative to r0), rows counted so far
        for (int k = 0; k < 12; ++k) {
            if (r0 + k + 2 > rows || k + 2 > panel_rows) break;
            const int a = rowptr[r0 + k], b = rowptr[r0 + k + 1], c2 = rowptr[r0 + k + 2];
            bool eq = b - a == c2 - b && b > a;
            if (eq) {
                bool differ = false;
                for (int e = lane; e < b - a; e += WAVE) differ |= colidx[a + e] != colidx[b + e];
                eq = __builtin_amdgcn_ballot_w64(differ) == 0ull;
            }
            if (start < 0) {
                if (!eq) start = k + 1, len = 1; // row k+1 opens a group
            } else if (eq) {
                ++len;
            } else {
                break; // the group is rows start .. start + len - 1
            }
            if (len > 9) break;
        }
        if (start >= 0 && len >= 3 && len <= 9 && len % 3 == 0) {
            shared = true;
            phase = (r0 + start) % 3;
            window_ok = false;
        }
    }
    // Matrix cores?  The fill of the 16 x 4 blocks a 16-row group touches, sampled on the panel's first 16 rows.  A
    // cheap filter first, on the panel's first row alone: the block fill cannot exceed a row's own fill of the 4-column
    // blocks it touches (entries / (4 x distinct blocks)), and that takes one pass over one row.
    // the longest of the panel's first 16 rows (the first row itself may be empty): its length and the number of
    // distinct 4-column blocks it touches, one pass over one row
    int head_adj = 0; // entries of that row whose column follows its predecessor's
    auto head_row = [&](int &len_out, int &runs_out) {
        const int r0 = p * panel_rows, r1 = min(min(r0 + 16, r0 + panel_rows), rows);
        int slen = 0, srow = r0;
        if (r0 + lane < r1) slen = rowptr[r0 + lane + 1] - rowptr[r0 + lane];
        int key = (slen << 4) | (15 - (lane & 15)); // longest row, lowest index first
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) key = max(key, __shfl_xor(key, m, WAVE));
        srow = r0 + 15 - (__builtin_amdgcn_readfirstlane(key) & 15);
        srow = min(srow, r1 - 1);
        const int a0 = rowptr[srow], b0 = rowptr[srow + 1];
        int runs = 0, adj = 0;
        for (int e = a0 + lane; e < b0; e += WAVE) {
            const int c = colidx[e], cp = e > a0 ? colidx[e - 1] : -8;
            runs += (c >> 2) != (cp >> 2);
            adj += c == cp + 1;
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            runs += __shfl_xor(runs, m, WAVE);
            adj += __shfl_xor(adj, m, WAVE);
        }
        len_out = b0 - a0;
        runs_out = runs;
        head_adj = adj;
    };
    // Rows whose columns come in runs (mesh numberings: nine consecutive columns per neighbour plane) are served well
    // out of the L2 by the direct kernels, so the LDS-tiled kernel pays later on them: grid-structured rows at 64
    // columns lose 16 % at 3.2 uses per loaded B row and win 12 % at 5.1 (uniform rows win from 2.75).  A panel under
    // the lowered bar whose head row fills 60 % of the 4-column blocks it touches needs 1.6 x the bar.
    if (window_ok && !strong) {
        int hl = 0, hr = 0;
        head_row(hl, hr);
        if (hl > 0 && (float)hl >= 0.6f * 4.0f * (float)hr && (float)nnz < 1.6f * min_density * (float)(last - first + 1))
            window_ok = false;
    }
    bool mfma = false;
    if (sane && mfma_min_fill <= 1.0f && last < 0x7fff0000) { // (the kernel's end-of-row sentinel is 0x7fffffff)
        const int r0 = p * panel_rows, r1 = min(min(r0 + 16, r0 + panel_rows), rows);
        int hl = 0, runs = 0;
        head_row(hl, runs);
        const int a0 = 0, b0 = hl;
        const bool candidate = b0 > a0 && (float)(b0 - a0) >= mfma_min_fill * 4.0f * (float)runs;
        const int e0 = rowptr[r0], e1 = rowptr[r1]; // the 16 rows' entries are one contiguous run of the CSR arrays
        const int blo = first >> 2;
        if (candidate && ((last >> 2) - blo) < MFMA_BITMAP_WORDS * 32) {
            for (int w = lane; w < MFMA_BITMAP_WORDS; w += WAVE) bitmap[w] = 0u;
            bool in_range = true;
            for (int e = e0 + lane; e < e1; e += 8 * WAVE) { // eight loads in flight per lane
                unsigned blk[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) blk[u] = (unsigned)((colidx[min(e + u * WAVE, e1 - 1)] >> 2) - blo);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (blk[u] < (unsigned)(MFMA_BITMAP_WORDS * 32)) atomicOr(&bitmap[blk[u] >> 5], 1u << (blk[u] & 31));
                    else in_range = false; // an unsorted row's entry outside [first, last]: not a candidate
                }
            }
            int nblk = 0, nchunk = 0; // blocks, and 64-column chunks (16 blocks) that hold one
            for (int w = lane; w < MFMA_BITMAP_WORDS; w += WAVE) {
                const unsigned bits = bitmap[w];
                nblk += __popc(bits);
                nchunk += ((bits & 0xffffu) != 0u) + ((bits >> 16) != 0u);
            }
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) {
                nblk += __shfl_xor(nblk, m, WAVE);
                nchunk += __shfl_xor(nchunk, m, WAVE);
            }
            // Panels the LDS-tiled kernel cannot take compete with the (slower) row-per-wave kernel, and the matrix-core
            // kernel pays a fixed price per 64-column chunk it visits.  Measured on such panels at N = 256: blocks
            // scattered one to a chunk (80 per row over a +-50 000 band) lose at any fill (50 %: 1.90 ms against 1.33
            // ms); with 4-5 blocks per chunk (grid-structured rows) 32 % fill loses by 8 %, 48 % wins by 23 %.  So:
            // at least two blocks per visited chunk, and from three on 0.84 of the fill threshold is enough.
            // ... and panels it can take compete with the LDS-tiled kernel, which round 3 made faster: block-structured
            // rows, N = 128 | 256, matrix cores against LDS tiles: 60 % fill 0.620 | 1.155 ms against 0.587 | 1.124, 70 %:
            // 0.560 | 1.052 against 0.560 | 1.092, 80 %: 0.534 | 0.998 against 0.597 | 1.112 -- the bar there is 0.68.
            float need = mfma_min_fill;
            bool dense_chunks = true;
            if (window_ok && mfma_min_fill > 0.0f) need = fmaxf(need, 0.68f);
            if (!window_ok && mfma_min_fill > 0.0f) {
                dense_chunks = nblk >= 2 * nchunk;
                if (nblk >= 3 * nchunk) need *= 0.84f;
            }
            mfma = dense_chunks && __builtin_amdgcn_ballot_w64(!in_range) == 0ull &&
                   (float)(e1 - e0) >= need * (float)(r1 - r0) * 4.0f * (float)nblk;
        }
    }
    // Which direct kernel (128+ staged columns; the vote below decides for the matrix)?  Four rows per wave on 64-column
    // tiles beat a row per wave on 128-column tiles on banded rows of any length (N = 256, 1 M rows, 32 / 64 / 100 per
    // row: 4.17 / 6.82 / 10.1 ms against 5.20 / 7.14 / 11.3), but not where a row's columns come in runs (Queen-like rows,
    // clusters of three: 4.78 against 3.88 ms), and four rows walk in step, so lengths far apart waste slots.
    bool wave_rows = false;
    if (sane && !window_ok && (merge_probe & 2)) {
        int hl = 0, hr = 0;
        head_row(hl, hr);
        const int prow = min(panel_rows, rows - p * panel_rows);
        wave_rows = (hl > 0 && 5 * head_adj >= 2 * hl) || (float)mlen * (float)prow > 2.0f * (float)nnz + 16.0f * (float)prow;
    }
    if (lane == 0) {
        info[p] = last >= first ? make_int2(first, last) : make_int2(1, 0);
        const int c = mfma ? (window_ok ? PANEL_MFMA_W : PANEL_MFMA_D) : (window_ok ? PANEL_WINDOW : PANEL_DIRECT);
        cls[p] = c | (shared ? PANEL_SHARED_ROWS : 0) | (phase << PANEL_PHASE_SHIFT) | (wave_rows ? PANEL_WAVE_ROWS : 0);
        // the middle panel's column span: the direct kernels take it as the band width of the matrix when they choose
        // their panel -> XCD map (one writer)
        if (p == npanels / 2) tail[TAIL_BAND] = last >= first ? last - first + 1 : 0;
        // "this call left work for the direct kernel": the launch's epoch (every writer stores the same value).  The
        // direct kernel leaves at once when the slot holds anything else -- an optimisation only: a stale or
        // accidental match merely sends it through its per-panel checks.
        if (c == PANEL_DIRECT) tail[TAIL_DIRECT_EPOCH] = epoch;
        if (c == PANEL_MFMA_D) tail[TAIL_MFMAD_EPOCH] = epoch;
        if (c == PANEL_MFMA_W || c == PANEL_MFMA_D) tail[TAIL_MFMA_EPOCH] = epoch;
    }
}
__global__ __launch_bounds__(256) void classify_panels_kernel(int rows, int cols, int npanels, int panel_rows,
                                                             const int *__restrict__ rowptr,
                                                             const int *__restrict__ colidx, int max_row_len,
                                                             float min_density, float min_rowlen, float mfma_min_fill, int merge_probe,
                                                             int *__restrict__ tail, int2 *__restrict__ info,
                                                             int *__restrict__ cls, int epoch)
{
    __shared__ unsigned bitmap[4][MFMA_BITMAP_WORDS];
    classify_panel(blockIdx.x * 4 + (threadIdx.x >> 6), rows, cols, npanels, panel_rows, rowptr, colidx, max_row_len,
                   min_density, min_rowlen, mfma_min_fill, merge_probe, tail, info, cls, epoch, bitmap[threadIdx.x >> 6]);
}
// Stage 1 and the panel classifier in one launch (the fused C-ABI entry: both depend only on the call's inputs, and
// the classifier's dependent loads hide behind the staging traffic): the first ceil(npanels / 4) workgroups
// (grid.y == 0 only) classify four panels each, the stage_blocks x grid.y behind them transpose B.
__global__ __launch_bounds__(256) void stage_classify_kernel(int64_t cols, int64_t n, const double *__restrict__ B,
                                                            int64_t ldb, double *__restrict__ Bt, int64_t ldbt,
                                                            int stage_blocks, int rows, int npanels, int panel_rows,
                                                            const int *__restrict__ rowptr,
                                                            const int *__restrict__ colidx, int max_row_len,
                                                            float min_density, float min_rowlen, float mfma_min_fill, int merge_probe,
                                                            int *__restrict__ tail, int2 *__restrict__ info,
                                                            int *__restrict__ cls, int epoch)
{
    // a workgroup either transposes (tile) or classifies (bitmaps): one LDS area serves both
    static_assert(sizeof(double) * 64 * (STAGE_K + 1) >= sizeof(unsigned) * 4 * MFMA_BITMAP_WORDS, "LDS area");
    __shared__ double tile[64][STAGE_K + 1];
    // the classifier's workgroups come first in the grid: their chain of dependent loads starts at once and ends
    // under the staging traffic (placed last they stuck out by ~3 us)
    const int cblocks = (npanels + 3) / 4;
    if ((int)blockIdx.x >= cblocks) {
        stage_tile(tile, (int64_t)((int)blockIdx.x - cblocks) * STAGE_K, (int64_t)blockIdx.y * 64, cols, n, B, ldb, Bt,
                   ldbt, tail, epoch);
    } else if (blockIdx.y == 0) {
        classify_panel((int)blockIdx.x * 4 + (threadIdx.x >> 6), rows, (int)cols, npanels, panel_rows, rowptr, colidx,
                       max_row_len, min_density, min_rowlen, mfma_min_fill, merge_probe, tail, info, cls, epoch,
                       reinterpret_cast<unsigned *>(&tile[0][0]) + (threadIdx.x >> 6) * MFMA_BITMAP_WORDS);
    }
}
// The same for a row block: the classifier rides with the column-range pass (both read only A); the staging launch
// that needs the range follows.
__global__ __launch_bounds__(256) void colrange_classify_kernel(int64_t nnz, int2 *__restrict__ part, int nparts, int rows,
                                                               int cols, int npanels, int panel_rows,
                                                               const int *__restrict__ rowptr,
                                                               const int *__restrict__ colidx, int max_row_len,
                                                               float min_density, float min_rowlen, float mfma_min_fill, int merge_probe,
                                                               int *__restrict__ tail, int2 *__restrict__ info,
                                                               int *__restrict__ cls, int epoch)
{
    __shared__ unsigned bitmaps[4 * MFMA_BITMAP_WORDS];
    const int cblocks = (npanels + 3) / 4;
    if ((int)blockIdx.x >= cblocks)
        colrange_part(nnz, colidx, part, (int)blockIdx.x - cblocks, nparts, reinterpret_cast<int2 *>(bitmaps));
    else
        classify_panel((int)blockIdx.x * 4 + (threadIdx.x >> 6), rows, cols, npanels, panel_rows, rowptr, colidx,
                       max_row_len, min_density, min_rowlen, mfma_min_fill, merge_probe, tail, info, cls, epoch,
                       bitmaps + (threadIdx.x >> 6) * MFMA_BITMAP_WORDS);
}
// ... and for narrow blocks (no matrix cores, no row-merging probe there)
template <int NC>
__global__ __launch_bounds__(256) void stage_classify_narrow_kernel(int64_t cols, int64_t n, const double *__restrict__ B,
                                                                   int64_t ldb, double *__restrict__ Bt, int rows,
                                                                   int npanels, int panel_rows,
                                                                   const int *__restrict__ rowptr,
                                                                   const int *__restrict__ colidx, int max_row_len,
                                                                   float min_density, float min_rowlen, int *__restrict__ tail,
                                                                   int2 *__restrict__ info, int *__restrict__ cls,
                                                                   int epoch)
{
    const int cblocks = (npanels + 3) / 4;
    if ((int)blockIdx.x >= cblocks)
        stage_rows_narrow<NC>((int64_t)((int)blockIdx.x - cblocks) * 256, cols, n, B, ldb, Bt, tail, epoch);
    else
        classify_panel((int)blockIdx.x * 4 + (threadIdx.x >> 6), rows, (int)cols, npanels, panel_rows, rowptr, colidx,
                       max_row_len, min_density, min_rowlen, 2.0f, 0, tail, info, cls, epoch, nullptr);
}
// The stage-2 kernels run one after the other, so a matrix whose panels split between the matrix-core kernel and the
// vector kernels pays for two half-empty launches (block-structured rows at the fill threshold, N = 128: 1.05 ms against
// 0.63 ms for either kernel alone).  One workgroup therefore settles the matter for the whole matrix before stage 2:
// unless at least three quarters of the non-empty panels qualified for the matrix cores, those that did go back to
// their vector kernel.  (Launched only where the matrix-core kernel can be chosen at all: 128+ staged columns.)
__global__ __launch_bounds__(1024) void mfma_vote_kernel(int npanels, int *__restrict__ tail, const int2 *__restrict__ info,
                                                       int *__restrict__ cls, int epoch, int mfma_forced)
{
    __shared__ int counts[4];
    if (threadIdx.x < 4) counts[threadIdx.x] = 0;
    __syncthreads();
    // (one workgroup, so the loads are what it waits for: eight panels' verdicts in flight per thread -- a million-row
    //  matrix has ten thousand panels, and the vote took 33 us with one at a time)
    constexpr int VU = 8;
    int mine = 0, all = 0, beyond = 0, groups = 0;
    for (int p0 = threadIdx.x; p0 < npanels; p0 += VU * 1024) {
        int2 sp[VU];
        int w[VU];
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int p = p0 + u * 1024;
            sp[u] = p < npanels ? info[p] : make_int2(1, 0);
            w[u] = p < npanels ? cls[p] : 0;
        }
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int c = w[u] & PANEL_CLASS_MASK;
            if (sp[u].x <= sp[u].y) {
                ++all;
                mine += c == PANEL_MFMA_W || c == PANEL_MFMA_D;
                if (c == PANEL_DIRECT || c == PANEL_MFMA_D) {
                    ++beyond;
                    groups += (w[u] & PANEL_SHARED_ROWS) != 0;
                }
            }
        }
    }
    // (wave sums first: 1024 lanes adding to one LDS word are served one by one -- eight such atomics per thread were
    //  20 us of this kernel whatever the panel count: 25.4 -> 5.4 us at 750 panels)
    auto tally = [&](int slot, int v) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, WAVE);
        if ((threadIdx.x & 63) == 0 && v != 0) atomicAdd(&counts[slot], v);
    };
    tally(0, mine);
    tally(1, all);
    tally(2, beyond);
    tally(3, groups);
    __syncthreads();
    const bool demote = !mfma_forced && counts[0] > 0 && 4 * counts[0] < 3 * counts[1];
    // Panels whose rows merge three at a time are better off with the row-merging kernel than with the matrix cores
    // (6-dof grid rows, N = 256: 1.86 ms against 2.81 ms): where half of the panels outside the LDS-tiled kernel's
    // reach show such groups, the matrix-core kernel's share of them goes to the direct path.
    const bool to_merge = !mfma_forced && counts[2] > 0 && 2 * counts[3] >= counts[2];
    __syncthreads();
    if (threadIdx.x < 4) counts[threadIdx.x] = 0;
    __syncthreads();
    // the direct kernel's panels after the vote: do their rows share column patterns?  (half of them: the row-merging
    // kernel takes all of the call's direct panels, otherwise the row-per-wave kernel does)
    int direct = 0, shared = 0, left = 0, by_wave = 0;
    bool any_direct = false;
    for (int p0 = threadIdx.x; p0 < npanels; p0 += VU * 1024) {
        int2 sp[VU];
        int w[VU];
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int p = p0 + u * 1024;
            sp[u] = p < npanels ? info[p] : make_int2(1, 0);
            w[u] = p < npanels ? cls[p] : PANEL_WINDOW;
        }
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int p = p0 + u * 1024;
            int c = w[u] & PANEL_CLASS_MASK;
            const int flag = w[u] & ~PANEL_CLASS_MASK; // shared-rows flag and group phase stay
            if (p < npanels && (demote || (to_merge && c == PANEL_MFMA_D))) {
                if (c == PANEL_MFMA_W) c = PANEL_WINDOW;
                if (c == PANEL_MFMA_D) {
                    c = PANEL_DIRECT;
                    any_direct = true;
                }
                cls[p] = c | flag;
            }
            left += c == PANEL_MFMA_W || c == PANEL_MFMA_D;
            if (c == PANEL_DIRECT && sp[u].x <= sp[u].y) {
                ++direct;
                shared += (flag & PANEL_SHARED_ROWS) != 0;
                by_wave += (flag & PANEL_WAVE_ROWS) != 0;
            }
        }
    }
    tally(2, direct);
    tally(3, shared);
    tally(0, left);
    tally(1, by_wave);
    if (any_direct) tail[TAIL_DIRECT_EPOCH] = epoch; // (every writer stores the same value)
    __syncthreads();
    if (threadIdx.x == 0) {
        if (counts[0] == 0) tail[TAIL_MFMA_EPOCH] = 0; // nothing for the matrix-core kernel (any more)
        if (counts[2] > 0 && 2 * counts[3] >= counts[2]) tail[TAIL_MERGE_EPOCH] = epoch;
        // ... else four rows per wave, unless half of them asked for a row per wave (classify_panel)
        else if (counts[2] > 0 && 2 * counts[1] < counts[2]) tail[TAIL_ROWS_EPOCH] = epoch;
    }
}
// no panel of this call is left to the direct kernel (every workgroup asks this first: two scalar loads, no per-panel work)
__device__ __forceinline__ bool nothing_direct(const int *__restrict__ tail, int epoch)
{
    if (tail[TAIL_DIRECT_EPOCH] == epoch) return false;
    return !(tail[TAIL_MFMAD_EPOCH] == epoch && tail[TAIL_NONFINITE] == tail[TAIL_STAGE_EPOCH]);
}
// which kernel computes panel p (cls == nullptr: no classifier ran, the direct kernel computes everything)
__device__ __forceinline__ bool b_nonfinite(const int *__restrict__ tail)
{
    return tail[TAIL_NONFINITE] == tail[TAIL_STAGE_EPOCH];
}
__device__ __forceinline__ bool owns_direct(const int *__restrict__ tail, const int *__restrict__ cls, int p)
{
    const int c = cls[p] & PANEL_CLASS_MASK;
    return c == PANEL_DIRECT || (c == PANEL_MFMA_D && b_nonfinite(tail));
}
__device__ __forceinline__ bool owns_window(const int *__restrict__ tail, const int *__restrict__ cls, int p)
{
    const int c = cls[p] & PANEL_CLASS_MASK;
    return c == PANEL_WINDOW || (c == PANEL_MFMA_W && b_nonfinite(tail));
}

typedef int sblas_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ sblas_rsrc_t make_rsrc(const void *p, unsigned stride, unsigned records)
{
    const unsigned long long a = (unsigned long long)p;
    sblas_rsrc_t r;
    r.x = (int)(unsigned)a;
    r.y = (int)(((unsigned)(a >> 32) & 0xffffu) | (stride << 16));
    r.z = (int)records;
    r.w = 0x00020000;
    return r;
}

// four steps (sixteen nonzeros) in one statement: all eight 16-byte LDS reads are in flight before the first FMA, so a
// typical visit (<= 16 nonzeros of a row in a 128-column tile) pays ONE LDS round trip.  Scratch v92..v127.
#define SBLAS_QSTEP4(K0, K1, K2, K3)                                                                                 \
    asm volatile("s_nop 1\n\t"                                                                                       \
                 "v_add_u32_dpp v92, %[co], %[lb] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"               \
                 "v_add_u32_dpp v93, %[co], %[lb] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"               \
                 "v_add_u32_dpp v94, %[co], %[lb] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"               \
                 "v_add_u32_dpp v95, %[co], %[lb] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"               \
                 "s_nop 0\n\t"                                                                                       \
                 "ds_read_b128 v[96:99], v92\n\t"                                                                    \
                 "ds_read_b128 v[100:103], v92 offset:256\n\t"                                                       \
                 "ds_read_b128 v[104:107], v93\n\t"                                                                  \
                 "ds_read_b128 v[108:111], v93 offset:256\n\t"                                                       \
                 "ds_read_b128 v[112:115], v94\n\t"                                                                  \
                 "ds_read_b128 v[116:119], v94 offset:256\n\t"                                                       \
                 "ds_read_b128 v[120:123], v95\n\t"                                                                  \
                 "ds_read_b128 v[124:127], v95 offset:256\n\t"                                                       \
                 "s_waitcnt lgkmcnt(7)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c0], %[gv], v[96:97] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"         \
                 "v_fmac_f64_dpp %[c1], %[gv], v[98:99] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"         \
                 "s_waitcnt lgkmcnt(6)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c2], %[gv], v[100:101] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_fmac_f64_dpp %[c3], %[gv], v[102:103] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "s_waitcnt lgkmcnt(5)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c0], %[gv], v[104:105] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_fmac_f64_dpp %[c1], %[gv], v[106:107] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "s_waitcnt lgkmcnt(4)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c2], %[gv], v[108:109] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_fmac_f64_dpp %[c3], %[gv], v[110:111] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "s_waitcnt lgkmcnt(3)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c0], %[gv], v[112:113] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_fmac_f64_dpp %[c1], %[gv], v[114:115] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "s_waitcnt lgkmcnt(2)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c2], %[gv], v[116:117] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_fmac_f64_dpp %[c3], %[gv], v[118:119] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "s_waitcnt lgkmcnt(1)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c0], %[gv], v[120:121] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_fmac_f64_dpp %[c1], %[gv], v[122:123] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c2], %[gv], v[124:125] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_fmac_f64_dpp %[c3], %[gv], v[126:127] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"       \
                 : [c0] "+v"(q0), [c1] "+v"(q1), [c2] "+v"(q2), [c3] "+v"(q3)                                        \
                 : [co] "v"(co), [lb] "v"(lb), [gv] "v"(gv)                                                          \
                 : "memory", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", \
                   "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",   \
                   "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127")

__device__ __forceinline__ void dma_rows_scalar(unsigned lds_addr, unsigned voff, const char *base)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base)
                 : "memory");
}
__device__ __forceinline__ void dma_rows_vector(unsigned lds_addr, const char *addr)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(addr)
                 : "memory");
}

static_assert(4 * 4 * 2 >= SPMM_MIN_PANEL_ROWS, "the workspace reserves one verdict per SPMM_MIN_PANEL_ROWS rows");
constexpr int W6_GMAX = 3;                   // groups of four rows per wave: 2 or 3 (template parameter)


// tile membership for one 16-entry half of the four windows; all per-lane (k = lane & 15 is the entry number inside
// the half, `rem` the entries left in the lane's matrix row counted from this half's first entry)
__device__ __forceinline__ void window_select6(int wc, double wv, int tile_lo, int rem, int k, unsigned zero_rel,
                                               unsigned &co, double &gv, unsigned long long &m)
{
    int glo, ghi;
    asm volatile("v_subrev_u32 %[co], %[tlo], %[wc]\n\t"
                 "v_cmp_gt_i32 %[m], %[rem], %[k]\n\t"
                 "v_cmp_gt_u32 vcc, %[tr], %[co]\n\t"
                 "v_lshlrev_b32 %[co], 9, %[co]\n\t"
                 "s_and_b64 vcc, vcc, %[m]\n\t"
                 "s_mov_b64 %[m], vcc\n\t"
                 "v_cndmask_b32 %[co], %[zr], %[co], vcc\n\t"
                 "v_cndmask_b32 %[glo], 0, %[vlo], vcc\n\t"
                 "v_cndmask_b32 %[ghi], 0, %[vhi], vcc"
                 : [co] "=&v"(co), [m] "=&s"(m), [glo] "=&v"(glo), [ghi] "=&v"(ghi)
                 : [tlo] "s"(tile_lo), [wc] "v"(wc), [rem] "v"(rem), [k] "v"(k), [zr] "v"(zero_rel),
                   [vlo] "v"(__double2loint(wv)), [vhi] "v"(__double2hiint(wv)), [tr] "n"(W2_ROWS)
                 : "vcc", "scc");
    gv = __hiloint2double(ghi, glo);
}
// the four 32-entry windows of a group: col_idx and val, entries 0-15 (A) and 16-31 (B) of every row
// NB the loads are asynchronous and the compiler does not know it: between this statement and the window_wait6 that retires
// them nothing may make it COPY the four registers (a copy made before the wait reads stale data).  The kernels keep the
// fetch as the last statement of a visit and the wait as the first of the next, which has held on every build so far;
// moving the fetch ahead of the narrow kernel's FMA block (round 3: to run its round trip under the LDS reads) made the
// register allocator put v_mov copies in front of the wait -- wrong results on the first parity case, not kept.
__device__ __forceinline__ void window_issue6(sblas_rsrc_t rc, sblas_rsrc_t rv, int idx, int &ca, double &va, int &cb,
                                              double &vb)
{
    const int idx2 = idx + 16;
    // (non-temporal loads here cost 19 %: every entry is used by two or three consecutive windows and those re-reads
    //  must hit)
    asm volatile("buffer_load_dword %0, %4, %6, 0 idxen\n\t"
                 "buffer_load_dwordx2 %1, %4, %7, 0 idxen\n\t"
                 "buffer_load_dword %2, %5, %6, 0 idxen\n\t"
                 "buffer_load_dwordx2 %3, %5, %7, 0 idxen"
                 : "=&v"(ca), "=&v"(va), "=&v"(cb), "=&v"(vb)
                 : "v"(idx), "v"(idx2), "s"(rc), "s"(rv)
                 : "memory");
}
template <int NEWER> __device__ __forceinline__ void window_wait6(int &ca, double &va, int &cb, double &vb)
{
    static_assert(NEWER == 0 || NEWER == 4 || NEWER == 8, "counts of the sixth-generation tile loop");
    if (NEWER == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(ca), "+v"(va), "+v"(cb), "+v"(vb)::"memory");
    else if (NEWER == 8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(ca), "+v"(va), "+v"(cb), "+v"(vb)::"memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(ca), "+v"(va), "+v"(cb), "+v"(vb)::"memory");
}
template <int N> __device__ __forceinline__ void vm_wait6()
{
    static_assert(N == 8 || N == 12, "counts of the sixth-generation tile loop");
    if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// NH = column halves per workgroup.  NH = 1: a 64-column tile of C (grid.y = ldbt / 64).  NH = 2 (128+ staged columns):
// the workgroup owns 128 columns and walks every 128-row tile of B twice -- columns 0-63, then 64-127, each through the
// same two LDS buffers -- and the second pass REUSES the first pass's selection (offsets, values, counts): the window
// wait, the two selections, the counting, the order check, the cursor update and the next window fetch are paid once per
// (rows, tile) visit instead of once per 64 columns (with grid.y = ldbt / 64 every 64-column slice re-streams A's windows
// and redoes all of it: N = 128 cost exactly twice N = 64).
template <int G, int NH>
__global__ __launch_bounds__(1024) void spmm_window6_kernel(
    int rows, int cols, int npanels, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, const int *__restrict__ tail, const int2 *__restrict__ info,
    const int *__restrict__ cls, int panel_rows, int nnz)
{
    constexpr int RW = 4 * G, RMAX = 16 * RW;
    static_assert(G == 2 || G == 3, "two or three groups per wave (the counted vmcnt waits are 4 G)");
    static_assert(NH == 1 || NH == 2, "one or two 64-column halves per workgroup");
    static_assert(64 * (RMAX + 1) <= 2 * W2_TILE, "C tile must fit in the (dead) B tile buffers");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *zero_row = smem + 2 * W2_TILE;
    int *sm_i = reinterpret_cast<int *>(smem + 2 * W2_TILE + 64); // [0] = bad

    const int panel = xcd_contiguous_panel(blockIdx.x, npanels);
    if (!owns_window(tail, cls, panel)) return; // another kernel owns this panel
    const int2 span = info[panel];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    const int row0 = panel * panel_rows;
    const int col0 = blockIdx.y * (64 * NH);
    const unsigned ld32 = (unsigned)ldbt;
    const int t_lo = span.x / W2_ROWS, t_hi = span.y / W2_ROWS;
    // waves 12-15 load the tiles, waves 0-11 consume (a wave that did both had its window loads retire behind its own
    // tile fetches -- vmcnt is in order -- and the kernel ran twice as long)
    const bool loader = wave >= 12;
    const bool active = !loader && wave * RW < panel_rows; // this wave has rows

    if (tid < 64) zero_row[tid] = 0.0;
    if (tid == 0) sm_i[0] = 0;

    double acc[NH][G][4];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][g][j] = 0.0;

    // ---- tile DMA: loader wave lw (= wave - 12) moves Bt rows 32 lw .. 32 lw + 31 of a tile: 16 instructions of two
    // rows each (lanes 0-31 the first, lanes 32-63 the second); tile row j lives at LDS byte j * 512.  Rows past the end
    // of B are clamped to row `cols`, the all-zero row of the workspace (scalar addresses when all 32 rows exist)
    const unsigned ldb8 = ld32 * 8u;
    const unsigned piece_off = (unsigned)(col0 + ((lane & 31) << 1)) * 8u; // bytes inside a Bt row (column half 0)
    const unsigned pair_off = (unsigned)(lane >> 5) * ldb8 + piece_off;
    const char *bt_bytes = reinterpret_cast<const char *>(Bt);
    int dummy = 0;
    // tile t, column half h -> buffer buf
    auto dma_part = [&](int t, int h, int buf) {
        const int lw = wave - 12;
        const int r0 = t * W2_ROWS + lw * W2_LROWS;
        const unsigned lds0 = (unsigned)wave_uniform((int)((unsigned)(uintptr_t)(smem + buf * W2_TILE) + (unsigned)lw * (unsigned)(W2_LROWS * 512)));
        const unsigned hoff = (unsigned)h * 512u;
        if (r0 + W2_LROWS - 1 <= cols) {
            const char *p = bt_bytes + (size_t)((unsigned)r0 * ldb8) + hoff;
#pragma unroll
            for (int i = 0; i < W2_LROWS / 2; ++i) dma_rows_scalar(lds0 + i * 1024u, pair_off, p + (size_t)(2u * i) * ldb8);
        } else {
#pragma unroll
            for (int i = 0; i < W2_LROWS / 2; ++i) {
                const unsigned brow = (unsigned)min(r0 + 2 * i + (lane >> 5), cols);
                dma_rows_vector(lds0 + i * 1024u, bt_bytes + (size_t)(brow * ldb8 + piece_off + hoff));
            }
        }
    };

    // the first tile does not depend on the row pointers: fetch it while they are on their way
    if (loader) dma_part(t_lo, 0, 0);

    const int k = lane & 15, q = lane >> 4;
    // per-lane row state, relative to the first nonzero of the wave's rows (base of the two descriptors)
    const int wrow = min(row0 + wave * RW, rows);
    const int wstart = wave_uniform(rowptr[wrow]);
    const sblas_rsrc_t rc = make_rsrc(colidx + wstart, 4u, (unsigned)(nnz - wstart));
    const sblas_rsrc_t rv = make_rsrc(val + wstart, 8u, (unsigned)(nnz - wstart));
    int cur[G], end[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int rr = wave * RW + 4 * g + q; // row inside the panel
        const int row = row0 + rr;
        cur[g] = end[g] = 0;
        if (rr < panel_rows && row < rows) {
            cur[g] = rowptr[row] - wstart;
            end[g] = rowptr[row + 1] - wstart;
        }
    }
    unsigned long long viol = 0ull; // lanes whose entry broke the "entries of a tile = window prefix" expectation
    int wca[G], wcb[G];
    double wva[G], wvb[G];
    // NH = 2: what the first half's visit selected, kept for the second half (one round; a visit that needed several
    // window rounds -- `replay` -- walks them again)
    unsigned s_coA[G], s_coB[G];
    double s_gvA[G], s_gvB[G];
    int s_take[G], s_mx[G];
    bool replay[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        s_coA[g] = s_coB[g] = 0u;
        s_gvA[g] = s_gvB[g] = 0.0;
        s_take[g] = s_mx[g] = 0;
        replay[g] = false;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) window_issue6(rc, rv, cur[g] + k, wca[g], wva[g], wcb[g], wvb[g]);
    // everything lands before the loop starts, so its counted waits (written for the steady state) hold from the
    // first tile on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads(); // P
    int step = 0;
    for (int t = t_lo; t <= t_hi; ++t) {
        const int tile_lo = t * W2_ROWS;
        auto half = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            const int cbuf = step & 1;
            // next (tile, half) (that buffer was last read before the previous barrier)
            if (loader) {
                if (h + 1 < NH) dma_part(t, h + 1, cbuf ^ 1);
                else if (t < t_hi) dma_part(t + 1, 0, cbuf ^ 1);
            }
            const unsigned tile_base = (unsigned)(uintptr_t)(smem + cbuf * W2_TILE);
            const unsigned lb = tile_base + (unsigned)k * 16u;
            const unsigned zero_rel = (unsigned)(uintptr_t)zero_row - tile_base;
            auto visit = [&](auto gc) {
                constexpr int g = decltype(gc)::value;
                // younger than this group's windows: NH = 1: the other groups' windows (4 each: every visit ends with
                // a fetch); NH = 2: the windows of the groups behind this one (the fetches of a tile are issued in its
                // last half, in group order)
                if (h == 0) window_wait6<NH == 1 ? 4 * (G - 1) : 4 * (G - 1 - g)>(wca[g], wva[g], wcb[g], wvb[g]);
                else if (replay[g]) window_wait6<0>(wca[g], wva[g], wcb[g], wvb[g]);
                double &q0 = acc[h][g][0], &q1 = acc[h][g][1], &q2 = acc[h][g][2], &q3 = acc[h][g][3];
                int lc = cur[g];
                bool full = h == 0 || replay[g];
                for (;;) {
                    unsigned coA, coB;
                    double gvA, gvB;
                    int take, mx;
                    if (full) {
                        const int rem = end[g] - lc;
                        unsigned long long mA, mB;
                        window_select6(wca[g], wva[g], tile_lo, rem, k, zero_rel, coA, gvA, mA);
                        window_select6(wcb[g], wvb[g], tile_lo, rem - 16, k, zero_rel, coB, gvB, mB);
                        // entries of this tile per matrix row (= per DPP row), as a per-lane value
                        const unsigned fa = q < 2 ? (unsigned)mA : (unsigned)(mA >> 32);
                        const unsigned fb = q < 2 ? (unsigned)mB : (unsigned)(mB >> 32);
                        const int sh = (q & 1) * 16;
                        take = __popc((fa >> sh) & 0xffffu) + __popc((fb >> sh) & 0xffffu);
                        // with ascending columns they are exactly the first `take` entries of the row's window
                        viol |= mA ^ __builtin_amdgcn_ballot_w64(k < take);
                        viol |= mB ^ __builtin_amdgcn_ballot_w64(k + 16 < take);
                        mx = max(max(__builtin_amdgcn_readlane(take, 0), __builtin_amdgcn_readlane(take, 16)),
                                 max(__builtin_amdgcn_readlane(take, 32), __builtin_amdgcn_readlane(take, 48)));
                        if (NH > 1 && h == 0) {
                            s_coA[g] = coA, s_coB[g] = coB, s_gvA[g] = gvA, s_gvB[g] = gvB, s_take[g] = take, s_mx[g] = mx;
                        }
                    } else {
                        coA = s_coA[g], coB = s_coB[g], gvA = s_gvA[g], gvB = s_gvB[g], take = s_take[g], mx = s_mx[g];
                    }
                    // (blocks of eight steps with the second half's LDS reads issued ahead of the first half's FMAs were
                    //  tried: no gain)
                    // The LDS-read / FMA blocks run at raised wave priority: a wave that has its operands selected gets the
                    // issue slots ahead of the waves still doing bookkeeping (step 0.2864 -> 0.2781 ms and 0.2690 -> 0.2563 ms on
                    // two boxes; priority 1, 2 or 3 alike; raised during the selection instead, on the loader waves, or on
                    // all consumers with the blocks one level higher: nothing)
                    asm volatile("s_setprio 1" ::: "memory");
                    {
                        const unsigned co = coA;
                        const double gv = gvA;
                        if (mx > 0) { SBLAS_QSTEP4(0, 1, 2, 3); }
                        if (mx > 4) { SBLAS_QSTEP4(4, 5, 6, 7); }
                        if (mx > 8) { SBLAS_QSTEP4(8, 9, 10, 11); }
                        if (mx > 12) { SBLAS_QSTEP4(12, 13, 14, 15); }
                    }
                    if (mx > 16) {
                        const unsigned co = coB;
                        const double gv = gvB;
                        SBLAS_QSTEP4(0, 1, 2, 3);
                        if (mx > 20) { SBLAS_QSTEP4(4, 5, 6, 7); }
                        if (mx > 24) { SBLAS_QSTEP4(8, 9, 10, 11); }
                        if (mx > 28) { SBLAS_QSTEP4(12, 13, 14, 15); }
                    }
                    asm volatile("s_setprio 0" ::: "memory");
                    lc += take;
                    // a row that used its whole window and has more: fetch the next windows now and go again (rare:
                    // more than 32 nonzeros of a row inside one 128-column tile)
                    const bool more = take >= 32 && lc < end[g];
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(more) == 0ull, 1)) break;
                    if (NH > 1 && h == 0) replay[g] = true;
                    window_issue6(rc, rv, lc + k, wca[g], wva[g], wcb[g], wvb[g]);
                    window_wait6<0>(wca[g], wva[g], wcb[g], wvb[g]); // drains the queue: later counted waits stay correct
                    full = true;
                }
                if (h == NH - 1) {
                    cur[g] = lc;
                    if (NH > 1) replay[g] = false;
                    // next tile's windows, into the registers this super-visit is done with
                    window_issue6(rc, rv, cur[g] + k, wca[g], wva[g], wcb[g], wvb[g]);
                } else if (replay[g]) {
                    // several rounds: the second half starts from the tile's first windows again
                    window_issue6(rc, rv, cur[g] + k, wca[g], wva[g], wcb[g], wvb[g]);
                }
            };
            if (active) {
                visit(std::integral_constant<int, 0>{});
                visit(std::integral_constant<int, 1>{});
                if constexpr (G > 2) visit(std::integral_constant<int, 2>{});
            }
            if (loader) asm volatile("s_waitcnt vmcnt(0)" : "+v"(dummy)::"memory"); // the tile has landed
            __syncthreads(); // E_t
            ++step;
        };
        half(std::integral_constant<int, 0>{});
        if constexpr (NH > 1) half(std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int g = 0; g < G; ++g) window_wait6<0>(wca[g], wva[g], wcb[g], wvb[g]); // retire the unused last fetches
    int bad = viol != 0ull ? 1 : 0;
#pragma unroll
    for (int g = 0; g < G; ++g)
        if (__builtin_amdgcn_ballot_w64(cur[g] < end[g]) != 0ull) bad = 1; // unconsumed nonzeros
    if (bad && lane == 0) atomicOr(&sm_i[0], 1);
    __syncthreads(); // V
    const bool fell_back = sm_i[0] != 0;
    if (tid == 0 && blockIdx.y == 0) atomicAdd(&g_panel_stats[fell_back ? 2 : 0], 1ull);

    // park the panel as [column][row] in the (now dead) tile buffers and write it back along rows, one 64-column half
    // at a time
    double *ctile = smem;
    const int nrows = min(panel_rows, rows - row0);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int colh = col0 + 64 * h;
        if (fell_back) {
            // recompute straight from L2, one column per lane: acc[h][g][j] <- row 4g+j of the wave
            const unsigned lane_off = (unsigned)(colh + lane);
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rr = wave * RW + 4 * g + j;
                    const int row = row0 + rr;
                    int a = 0, b = 0;
                    if (rr < panel_rows && row < rows) {
                        a = wave_uniform(rowptr[row]);
                        b = wave_uniform(rowptr[row + 1]);
                    }
                    acc[h][g][j] = row_direct(colidx, val, Bt, ld32, lane_off, lane, a, b);
                }
        }
        if (h > 0) __syncthreads(); // the previous half has been written back
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (fell_back) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ctile[lane * (RMAX + 1) + wave * RW + 4 * g + j] = acc[h][g][j];
            } else {
                const int rr = wave * RW + 4 * g + q;
                ctile[(2 * k) * (RMAX + 1) + rr] = acc[h][g][0];
                ctile[(2 * k + 1) * (RMAX + 1) + rr] = acc[h][g][1];
                ctile[(32 + 2 * k) * (RMAX + 1) + rr] = acc[h][g][2];
                ctile[(33 + 2 * k) * (RMAX + 1) + rr] = acc[h][g][3];
            }
        }
        __syncthreads(); // F
        const int ncols = min(64, n - colh);
        // (fetching the old C values in the prologue, to take their HBM latency out of the epilogue, did not pay: the
        //  registers they hold across the tile loop spill)
        for (int idx = tid; idx < 64 * panel_rows; idx += 1024) {
            const int r = idx % panel_rows, j = idx / panel_rows;
            if (r < nrows && j < ncols) {
                double *dst = C + (int64_t)(colh + j) * ldc + (row0 + r);
                const double sres = alpha * ctile[j * (RMAX + 1) + r];
                *dst = (beta == 0.0) ? sres : fma(beta, *dst, sres);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 2, LDS-tiled form for NARROW dense blocks (ldbt = NC in {8, 16, 32}: what method 1 hands a GPU when N = 64 is
// split over 8 / 4 / 2 of them, matrix.h:554-568).
//
// Same skeleton as spmm_window6_kernel (row panel x LDS tiles of Bt, four LDS-DMA loader waves, one DPP row per matrix
// row, 32-entry windows of A fetched a tile ahead with counted vmcnt), but the work inside a visit is turned round: a
// Bt row is only NC * 8 = 64 ... 256 bytes, so instead of broadcasting entry k of a row to sixteen lanes that each
// hold four columns, EVERY LANE COMPUTES ITS OWN ENTRY: lane k of DPP row q reads the whole Bt row of entry k of
// matrix row q (NC / 2 ds_read_b128) and adds its NC products to NC accumulators of its own.  A visit is then
// two selections, NC / 2 LDS reads and NC FMAs per half window -- no per-entry broadcast steps, no step-count chain --
// and the partial sums of a matrix row are added up once per panel (LDS atomics into the parked C tile).
// Tiles are 256 Bt rows (a row of the bench matrix has ~25 entries in one: most of a 32-entry window).  Bank conflicts
// of the gather: a lane reads its row's 16-byte pieces in rotated order (start = its position in the sixteen-lane
// group that one LDS cycle serves), and a tile row may be stored CP times over so that lanes of a group never share a
// bank quad; measured, the kernel is bound by instruction issue, not by the LDS port -- one copy (the least tile DMA)
// is as fast as two or four (N = 8: 0.121 / 0.122 / 0.129 ms per step, N = 16: 0.144 / 0.143), so CP = 1 ships.
// From 16 columns on two lanes may share an entry (LPE = 2: lanes k and k ^ 8 of a DPP row take the two halves of the
// row): half the accumulators per lane, so a wave carries a group of rows more (taller panels: less tile DMA and
// fewer prologues per row) -- N = 32: 0.206 -> 0.180 ms, N = 16: 0.139 -> 0.136.
// Measured on the bench matrix (72 000 rows, 399 per row), whole step: N = 8 0.117-0.121 ms, N = 16 0.136-0.142 ms,
// N = 32 0.180 ms (64-column path of round 2: 0.265 / 0.238 / 0.245 ms).
// ---------------------------------------------------------------------------------------------
constexpr int WL_TR = 256;          // Bt rows per LDS tile
template <int NC, int CP> struct WlGeom {
    static constexpr int P = NC / 2;                 // 16-byte pieces per Bt row
    static constexpr unsigned ROWB = NC * 8;         // bytes of a Bt row
    static constexpr unsigned ROW = CP * ROWB;       // LDS bytes per tile row: CP copies of the Bt row
    static constexpr int SHIFT = ROW == 64 ? 6 : ROW == 128 ? 7 : 8;
    static constexpr unsigned TILE = WL_TR * ROW;
    static constexpr size_t LDS_BYTES = 2 * (size_t)TILE + 512 + 64;
    static constexpr int LPR = CP * P;               // lanes of a DMA instruction per tile row
    static constexpr int RPI = 64 / LPR;             // tile rows per DMA instruction (1 KiB)
    static_assert(ROW == 64 || ROW == 128 || ROW == 256, "copies x row bytes");
    static_assert(CP * P <= 16, "a lane group has sixteen lanes");
};

// position of a lane inside the group of sixteen lanes that one LDS cycle of a ds_read_b128 serves
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32: MI355X_MICROARCH.md, LDS)
__device__ __forceinline__ int b128_group_pos(int lane)
{
    const int l = lane & 31;
    if (l < 4) return l;            // group A
    if (l < 12) return l - 4;       // group B
    if (l < 16) return l - 8;       // group A: 12-15 -> 4-7
    if (l < 20) return l - 8;       // group B: 16-19 -> 8-11
    if (l < 28) return l - 12;      // group A: 20-27 -> 8-15
    return l - 16;                  // group B: 28-31 -> 12-15
}

template <int SHIFT>
__device__ __forceinline__ void lanes_select(int wc, double wv, int tile_lo, int rem, int k, unsigned zero_rel,
                                             unsigned &co, double &gv, unsigned long long &m)
{
    int glo, ghi;
    asm volatile("v_subrev_u32 %[co], %[tlo], %[wc]\n\t"
                 "v_cmp_gt_i32 %[m], %[rem], %[k]\n\t"
                 "v_cmp_gt_u32 vcc, 0x100, %[co]\n\t"
                 "v_lshlrev_b32 %[co], %[sh], %[co]\n\t"
                 "s_and_b64 vcc, vcc, %[m]\n\t"
                 "s_mov_b64 %[m], vcc\n\t"
                 "v_cndmask_b32 %[co], %[zr], %[co], vcc\n\t"
                 "v_cndmask_b32 %[glo], 0, %[vlo], vcc\n\t"
                 "v_cndmask_b32 %[ghi], 0, %[vhi], vcc"
                 : [co] "=&v"(co), [m] "=&s"(m), [glo] "=&v"(glo), [ghi] "=&v"(ghi)
                 : [tlo] "s"(tile_lo), [wc] "v"(wc), [rem] "v"(rem), [k] "v"(k), [zr] "v"(zero_rel),
                   [vlo] "v"(__double2loint(wv)), [vhi] "v"(__double2hiint(wv)), [sh] "n"(SHIFT)
                 : "vcc", "scc");
    gv = __hiloint2double(ghi, glo);
    static_assert(WL_TR == 256, "the literal of the selection");
}
typedef double sblas_d2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) sblas_d2 *sblas_lds_d2;
// The products of one half window: lane-own entry x its Bt row.  Lane l reads the row's pieces in the order
// rot, rot + 1, ..., P - 1, 0, ..., rot - 1 (rot = its position in the LDS lane group), from copy `copy` of the row: the
// sixteen lanes of a group then sit on sixteen different bank quads whatever rows they read.  acc[j] therefore holds
// the column pair (j + rot) mod P.  wrap[j] = lanes whose j-th piece is past the end of the row (they read from
// `hi` = lo - row bytes).
// PL = pieces a lane reads of its entry's row: all P of them, or P / 2 when two lanes share an entry (NC = 32: lane k and
// lane k ^ 8 of a DPP row take the two 128-byte halves of the row).
template <int PL>
__device__ __forceinline__ void lanes_fma(unsigned lo, double gv, const unsigned long long (&wrap)[PL], sblas_d2 (&acc)[PL])
{
    constexpr int CH = PL < 4 ? PL : 4;
    const unsigned hi = lo - (unsigned)(PL * 16);
#pragma unroll
    for (int j0 = 0; j0 < PL; j0 += CH) {
        sblas_d2 b[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int j = j0 + u;
            unsigned a = lo;
            if (j > 0) asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(a) : "v"(lo), "v"(hi), "s"(wrap[j]));
            b[u] = ((sblas_lds_d2)(uintptr_t)a)[j];
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            acc[j0 + u].x = fma(gv, b[u].x, acc[j0 + u].x);
            acc[j0 + u].y = fma(gv, b[u].y, acc[j0 + u].y);
        }
    }
}
// value of lane (k ^ 8) of the same DPP row
__device__ __forceinline__ unsigned dpp_ror8(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false); }
__device__ __forceinline__ double dpp_ror8(double x)
{
    return __hiloint2double((int)dpp_ror8((unsigned)__double2hiint(x)), (int)dpp_ror8((unsigned)__double2loint(x)));
}
// one row of a narrow block, straight from Bt in global memory (the per-panel fallback; kept out of line and eight
// columns at a time so that it does not set the kernel's register count): every lane takes every 64th entry, the 64
// partial sums are folded at the end.  Lane j < NC returns column j.
template <int NC>
__device__ __noinline__ double row_direct_narrow(const int *__restrict__ colidx, const double *__restrict__ val,
                                                 const double *__restrict__ Bt, int lane, int p0, int p1)
{
    double mine = 0.0;
    for (int j0 = 0; j0 < NC; j0 += 8) {
        double a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = 0.0;
        for (int p = p0 + lane; p < p1; p += WAVE) {
            const double v = val[p];
            const double *br = Bt + (size_t)colidx[p] * NC + j0;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = fma(v, br[j], a[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double s = a[j];
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, WAVE);
            if (lane == j0 + j) mine = s;
        }
    }
    return mine;
}

// 8 columns, one group of rows per wave: 63 registers and 33 KB of LDS, so TWO workgroups share a CU -- one's prologue and
// epilogue (a fifth of a panel's time at this width) run under the other's tile loop, eight waves per SIMD hide the LDS
// and window round trips: 48-row panels this way 0.117-0.118 ms per step against 0.124 with one workgroup of 144-row
// panels per CU (and 0.134 with one workgroup of 48-row panels).  At 16 columns (two lanes per entry, spills at 64
// registers) the same bought 1 %: not kept.
template <int NC, int CP, int G, int LPE>
__global__ __launch_bounds__(1024, (NC == 8 && CP == 1 && G == 1) ? 8 : 4) void spmm_lanes_kernel(
    int rows, int cols, int npanels, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, const int *__restrict__ tail, const int2 *__restrict__ info,
    const int *__restrict__ cls, int panel_rows, int nnz)
{
    using Geo = WlGeom<NC, CP>;
    constexpr int P = Geo::P;
    constexpr int PL = P / LPE; // pieces per lane
    constexpr unsigned WL_TILE = Geo::TILE;
    constexpr int RW = 4 * G, RMAX = 12 * RW;
    static_assert(NC == 8 || NC == 16 || NC == 32, "8, 16 or 32 dense columns");
    static_assert(LPE == 1 || (LPE == 2 && NC >= 16 && CP == 1), "two lanes per entry: 16 or 32 columns, one copy");
    static_assert(G >= 1 && G <= 3, "one to three groups of four rows per wave (counted vmcnt waits: 4 G)");
    static_assert((size_t)NC * (RMAX + 1) * sizeof(double) <= 2 * (size_t)WL_TILE, "C tile must fit in the (dead) B tiles");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    char *sm = reinterpret_cast<char *>(smem);
    double *zero_row = reinterpret_cast<double *>(sm + 2 * WL_TILE); // 512 bytes
    int *sm_i = reinterpret_cast<int *>(sm + 2 * WL_TILE + 512);     // [0] = bad

    const int panel = xcd_contiguous_panel(blockIdx.x, npanels);
    if (!owns_window(tail, cls, panel)) return; // another kernel owns this panel
    const int2 span = info[panel];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    const int row0 = panel * panel_rows;
    const int t_lo = span.x / WL_TR, t_hi = span.y / WL_TR;
    const bool loader = wave >= 12;
    const bool active = !loader && wave * RW < panel_rows;

    if (tid < 64) zero_row[tid] = 0.0;
    if (tid == 0) sm_i[0] = 0;

    sblas_d2 acc[G][PL];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < PL; ++j) acc[g][j] = sblas_d2{0.0, 0.0};

    // ---- tile DMA: loader wave lw moves Bt rows 64 lw .. 64 lw + 63 of a tile, RPI rows (1 KiB of LDS) per
    // instruction: lane l fills slot l % LPR of row l / LPR with piece (l % LPR) mod P of that Bt row -- the row is stored
    // CP times over.  Rows past the end of B are clamped to row `cols`, the all-zero row of the workspace.
    const char *bt_bytes = reinterpret_cast<const char *>(Bt);
    const unsigned dma_piece = (unsigned)((lane % Geo::LPR) & (P - 1)) * 16u;
    const unsigned dma_voff = (unsigned)(lane / Geo::LPR) * Geo::ROWB + dma_piece;
    int dummy = 0;
    auto dma_tile = [&](int t, int buf) {
        const int lw = wave - 12;
        const int r0 = t * WL_TR + lw * 64;
        const unsigned lds0 = (unsigned)(uintptr_t)(sm + (size_t)buf * WL_TILE) + (unsigned)lw * (64u * Geo::ROW);
        constexpr int NI = 64 / Geo::RPI;
        if (r0 + 63 <= cols) {
            const char *p = bt_bytes + (size_t)r0 * Geo::ROWB;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                dma_rows_scalar(lds0 + i * 1024u, dma_voff, p + (size_t)(Geo::RPI * i) * Geo::ROWB);
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const unsigned brow = (unsigned)min(r0 + Geo::RPI * i + lane / Geo::LPR, cols);
                dma_rows_vector(lds0 + i * 1024u, bt_bytes + (size_t)brow * Geo::ROWB + dma_piece);
            }
        }
    };
    if (loader) dma_tile(t_lo, 0);

    const int k = lane & 15, q = lane >> 4;
    // this lane's place in its LDS lane group: which copy of a tile row it reads and where in the row it starts
    // LPE = 2: lanes k and k ^ 8 of a DPP row share an entry; lane k reads half k >> 3 of the row, starting at piece k & 7
    // of that half (with the lane groups of ds_read_b128 -- b128_group_pos -- the sixteen lanes of a group again sit on
    // sixteen different bank quads)
    const int gpos = b128_group_pos(lane);
    const int rot = LPE == 2 ? (k & (PL - 1)) : (gpos & (P - 1)), copy = LPE == 2 ? 0 : (gpos / P) % CP;
    const int half = LPE == 2 ? (k >> 3) : 0;
    const unsigned lane_off = (unsigned)copy * Geo::ROWB + (unsigned)half * (PL * 16u) + (unsigned)rot * 16u;
    unsigned long long wrap[PL];
#pragma unroll
    for (int j = 0; j < PL; ++j) wrap[j] = __builtin_amdgcn_ballot_w64(j + rot >= PL);
    const bool upper = k >= 8;

    const int wrow = min(row0 + wave * RW, rows);
    const int wstart = wave_uniform(rowptr[wrow]);
    const sblas_rsrc_t rc = make_rsrc(colidx + wstart, 4u, (unsigned)(nnz - wstart));
    const sblas_rsrc_t rv = make_rsrc(val + wstart, 8u, (unsigned)(nnz - wstart));
    int cur[G], end[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int rr = wave * RW + 4 * g + q;
        const int row = row0 + rr;
        cur[g] = end[g] = 0;
        if (rr < panel_rows && row < rows) {
            cur[g] = rowptr[row] - wstart;
            end[g] = rowptr[row + 1] - wstart;
        }
    }
    unsigned long long viol = 0ull;
    int wca[G], wcb[G];
    double wva[G], wvb[G];
#pragma unroll
    for (int g = 0; g < G; ++g) window_issue6(rc, rv, cur[g] + k, wca[g], wva[g], wcb[g], wvb[g]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads(); // P
    for (int t = t_lo; t <= t_hi; ++t) {
        const int cbuf = (t - t_lo) & 1;
        if (loader && t < t_hi) dma_tile(t + 1, cbuf ^ 1);
        const int tile_lo = t * WL_TR;
        const unsigned tile_base = (unsigned)(uintptr_t)(sm + (size_t)cbuf * WL_TILE);
        const unsigned lb = tile_base + lane_off;
        const unsigned zero_rel = (unsigned)(uintptr_t)zero_row - tile_base;
        auto visit = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            window_wait6<4 * (G - 1)>(wca[g], wva[g], wcb[g], wvb[g]);
            for (;;) {
                const int rem = end[g] - cur[g];
                unsigned coA, coB;
                double gvA, gvB;
                unsigned long long mA, mB;
                lanes_select<Geo::SHIFT>(wca[g], wva[g], tile_lo, rem, k, zero_rel, coA, gvA, mA);
                lanes_select<Geo::SHIFT>(wcb[g], wvb[g], tile_lo, rem - 16, k, zero_rel, coB, gvB, mB);
                const unsigned fa = q < 2 ? (unsigned)mA : (unsigned)(mA >> 32);
                const unsigned fb = q < 2 ? (unsigned)mB : (unsigned)(mB >> 32);
                const int sh = (q & 1) * 16;
                const int take = __popc((fa >> sh) & 0xffffu) + __popc((fb >> sh) & 0xffffu);
                // with ascending columns the entries of this tile are exactly the first `take` entries of the window
                viol |= mA ^ __builtin_amdgcn_ballot_w64(k < take);
                viol |= mB ^ __builtin_amdgcn_ballot_w64(k + 16 < take);
                asm volatile("s_setprio 1" ::: "memory");
                if constexpr (LPE == 1) {
                    if (mA != 0ull) lanes_fma<PL>(lb + coA, gvA, wrap, acc[g]);
                    asm volatile("" ::: "memory");
                    if (mB != 0ull) lanes_fma<PL>(lb + coB, gvB, wrap, acc[g]);
                } else {
                    // four sets of eight entries per row: entry (k & 7) of the set goes to lanes k and k ^ 8
                    const unsigned long long LOW8 = 0x00ff00ff00ff00ffull;
                    const unsigned coAr = dpp_ror8(coA), coBr = dpp_ror8(coB);
                    const double gvAr = dpp_ror8(gvA), gvBr = dpp_ror8(gvB);
                    if ((mA & LOW8) != 0ull) lanes_fma<PL>(lb + (upper ? coAr : coA), upper ? gvAr : gvA, wrap, acc[g]);
                    asm volatile("" ::: "memory");
                    if ((mA & ~LOW8) != 0ull) lanes_fma<PL>(lb + (upper ? coA : coAr), upper ? gvA : gvAr, wrap, acc[g]);
                    asm volatile("" ::: "memory");
                    if ((mB & LOW8) != 0ull) lanes_fma<PL>(lb + (upper ? coBr : coB), upper ? gvBr : gvB, wrap, acc[g]);
                    asm volatile("" ::: "memory");
                    if ((mB & ~LOW8) != 0ull) lanes_fma<PL>(lb + (upper ? coB : coBr), upper ? gvB : gvBr, wrap, acc[g]);
                }
                asm volatile("s_setprio 0" ::: "memory");
                cur[g] += take;
                const bool more = take >= 32 && cur[g] < end[g];
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(more) == 0ull, 1)) break;
                window_issue6(rc, rv, cur[g] + k, wca[g], wva[g], wcb[g], wvb[g]);
                window_wait6<0>(wca[g], wva[g], wcb[g], wvb[g]);
            }
            window_issue6(rc, rv, cur[g] + k, wca[g], wva[g], wcb[g], wvb[g]);
        };
        if (active) {
            visit(std::integral_constant<int, 0>{});
            if constexpr (G > 1) visit(std::integral_constant<int, 1>{});
            if constexpr (G > 2) visit(std::integral_constant<int, 2>{});
        }
        if (loader) asm volatile("s_waitcnt vmcnt(0)" : "+v"(dummy)::"memory"); // the tile has landed
        __syncthreads(); // E_t
    }
#pragma unroll
    for (int g = 0; g < G; ++g) window_wait6<0>(wca[g], wva[g], wcb[g], wvb[g]);
    int bad = viol != 0ull ? 1 : 0;
#pragma unroll
    for (int g = 0; g < G; ++g)
        if (__builtin_amdgcn_ballot_w64(cur[g] < end[g]) != 0ull) bad = 1;
    if (bad && lane == 0) atomicOr(&sm_i[0], 1);
    // the C tile [NC][RMAX + 1] takes the place of the (now dead) B tiles; the lanes' partial sums are ADDED into it
    double *ctile = smem;
    __syncthreads(); // V: every wave is done with the tiles
    const bool fell_back = sm_i[0] != 0;
    if (tid == 0) atomicAdd(&g_panel_stats[fell_back ? 2 : 0], 1ull);
    if (!fell_back) {
        for (int i = tid; i < NC * (RMAX + 1); i += 1024) ctile[i] = 0.0;
        __syncthreads();
        if (!loader) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int rr = wave * RW + 4 * g + q;
#pragma unroll
                for (int j = 0; j < PL; ++j) {
                    const int pc = half * PL + ((j + rot) & (PL - 1)); // the column pair acc[g][j] holds
                    unsafeAtomicAdd(&ctile[(2 * pc) * (RMAX + 1) + rr], acc[g][j].x);
                    unsafeAtomicAdd(&ctile[(2 * pc + 1) * (RMAX + 1) + rr], acc[g][j].y);
                }
            }
        }
    } else if (!loader) {
        for (int j = 0; j < RW; ++j) { // (wave-uniform)
            const int rr = wave * RW + j;
            const int row = row0 + rr;
            int a = 0, b = 0;
            if (rr < panel_rows && row < rows) {
                a = wave_uniform(rowptr[row]);
                b = wave_uniform(rowptr[row + 1]);
            }
            const double s = row_direct_narrow<NC>(colidx, val, Bt, lane, a, b);
            if (lane < NC) ctile[lane * (RMAX + 1) + rr] = s;
        }
    }
    __syncthreads(); // F
    const int nrows = min(panel_rows, rows - row0);
    const int ncols = min(NC, n);
    for (int idx = tid; idx < NC * panel_rows; idx += 1024) {
        const int r = idx % panel_rows, j = idx / panel_rows;
        if (r < nrows && j < ncols) {
            double *dst = C + (int64_t)j * ldc + (row0 + r);
            const double sres = alpha * ctile[j * (RMAX + 1) + r];
            *dst = (beta == 0.0) ? sres : fma(beta, *dst, sres);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 2, direct form with DPP broadcast (any matrix; no assumption on column order or locality).
//
// A wave owns a row and a 128-column tile of C: every lane holds TWO adjacent columns, so a Bt row segment
// is fetched with one 16-byte load per lane (a wave64 load instruction occupies the address unit for 16
// cycles whatever its width, so 8-byte loads halve the vector-memory data rate).  For a 64-column tile
// (HALF) the two 32-lane halves work on two different nonzeros of the same row at once and are added at
// the end.  The 64-nonzero register chunk is redistributed with ds_bpermute so that every 16-lane DPP row
// holds 16 consecutive nonzeros (rows 0-1: nonzeros e..e+15, rows 2-3: e+16..e+31 when HALF, else all four
// rows the same 16); `row_newbcast:k` then feeds nonzero k to every lane of a row at full VALU rate:
//     off_k = row_newbcast:k(col*ldbt*8) + lane_byte      (v_add_u32_dpp)
//     b     = 16 bytes at Bt + off_k                       (global_load_dwordx4, scalar base + vector offset)
//     acc0 += row_newbcast:k(val) * b.x ; acc1 += ... b.y  (v_fmac_f64_dpp)
// Slots past the end of the row carry value 0 and the offset of the all-zero row Bt[cols].
// ---------------------------------------------------------------------------------------------
#define SBLAS_DPP_OFF4(K0, K1, K2, K3, O0, O1, O2, O3)                                                               \
    asm volatile("s_nop 1\n\t"                                                                                       \
                 "v_add_u32_dpp %[o0], %[co], %[lb] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "v_add_u32_dpp %[o1], %[co], %[lb] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "v_add_u32_dpp %[o2], %[co], %[lb] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "v_add_u32_dpp %[o3], %[co], %[lb] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"             \
                 : [o0] "=&v"(O0), [o1] "=&v"(O1), [o2] "=&v"(O2), [o3] "=&v"(O3)                                    \
                 : [co] "v"(co), [lb] "v"(lb))

#define SBLAS_DPP_FMA4x2(K0, K1, K2, K3, B0, B1, B2, B3)                                                             \
    asm volatile("s_nop 1\n\t"                                                                                       \
                 "v_fmac_f64_dpp %[c0], %[gv], %[x0] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "v_fmac_f64_dpp %[c1], %[gv], %[y0] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "v_fmac_f64_dpp %[c0], %[gv], %[x1] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "v_fmac_f64_dpp %[c1], %[gv], %[y1] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "v_fmac_f64_dpp %[c0], %[gv], %[x2] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "v_fmac_f64_dpp %[c1], %[gv], %[y2] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "v_fmac_f64_dpp %[c0], %[gv], %[x3] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "v_fmac_f64_dpp %[c1], %[gv], %[y3] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"            \
                 : [c0] "+v"(acc0), [c1] "+v"(acc1)                                                                  \
                 : [gv] "v"(gv), [x0] "v"(B0.x), [y0] "v"(B0.y), [x1] "v"(B1.x), [y1] "v"(B1.y), [x2] "v"(B2.x),    \
                   [y2] "v"(B2.y), [x3] "v"(B3.x), [y3] "v"(B3.y))

// GROUPS = 1: 128-column tile, one nonzero per instruction; 2: 64 columns, two nonzeros; 4: 32 columns, four (one per
// DPP row) -- n <= 32 runs on the 64-column staging copy and reads the first half of every Bt row
template <int GROUPS>
__global__ __launch_bounds__(WIDE_WAVES * 64) void spmm_direct_dpp_kernel(
    int rows, int cols, int npanels, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, const int *__restrict__ tail, const int *__restrict__ cls,
    int info_panel_rows, int interleave, int epoch, int long_min)
{
    static_assert(GROUPS == 1 || GROUPS == 2 || GROUPS == 4, "lane groups of 64, 32 or 16 lanes");
    constexpr int TILE_COLS = 128 / GROUPS;
    constexpr int PER_STEP = 16 * GROUPS; // nonzeros handled by one 16-slot sweep
    constexpr int GLANES = 64 / GROUPS;   // lanes that share a nonzero
    __shared__ double ctile[TILE_COLS][WIDE_PANEL + 1];
    // every panel windowed (the bench matrix): one scalar load of one shared address and out, instead of two
    // dependent loads per workgroup (4500 workgroups of early exits took 16 us of a 340 us step)
    if (cls != nullptr && nothing_direct(tail, epoch)) return;
    // the row-merging kernel's call, or the four-rows-per-wave kernel's
    if (GROUPS == 1 && cls != nullptr && (tail[TAIL_MERGE_EPOCH] == epoch || tail[TAIL_ROWS_EPOCH] == epoch)) return;
    const int lane = threadIdx.x & 63;
    const int wave = wave_uniform(threadIdx.x >> 6);
    // interleave: neighbouring panels on different XCDs, so that the whole chip sweeps one band of B at a time (wide
    // bands: the band must fit the Infinity Cache once, not once per XCD); otherwise one contiguous range per XCD
    // (`interleave` < 0: decide from the column span the classifier recorded -- a band of B rows wider than 16 MB)
    if (interleave < 0) {
        interleave = 0;
        if (cls != nullptr) { // one value for every workgroup: the middle panel's span, left by the classifier
            const int band = tail[TAIL_BAND];
            interleave = (long long)band * (TILE_COLS * 8) > (16ll << 20);
        }
    }
    // (a persistent form -- a few workgroups per CU walking the panels, two scalar loads per skipped panel -- was tried
    //  to make the all-windowed case cheaper: the direct case lost 10-15 % to the static assignment, no gain overall)
    const int row0 = (interleave ? (int)blockIdx.x : xcd_contiguous_panel(blockIdx.x, npanels)) * WIDE_PANEL;
    const int col0 = blockIdx.y * TILE_COLS;
    const int row = row0 + wave;
    // rows of panels that the windowed kernel owns are skipped (wave-uniform: one row per wave)
    bool mine = row < rows;
    if (cls && mine) {
        mine = owns_direct(tail, cls, row / info_panel_rows);
        if (mine && lane == 0 && blockIdx.y == 0 && row % info_panel_rows == 0) atomicAdd(&g_panel_stats[1], 1ull);
    }
    const int sub = lane & 15;
    const int half = lane / GLANES; // which of the GROUPS nonzeros of a step this lane works on
    const unsigned ldb8 = (unsigned)ldbt * 8u;                                       // bytes per Bt row
    const unsigned lb = (unsigned)(col0 * 8) + (unsigned)(lane % GLANES) * 16u; // this lane's 2 columns
    const unsigned zero_off = (unsigned)cols * ldb8;                                 // Bt[cols][*] == 0
    const char *__restrict__ bt_bytes = reinterpret_cast<const char *>(Bt);

    double acc0 = 0.0, acc1 = 0.0;
    // entries [p0, p1) of one row into (acc0, acc1)
    auto sweep = [&](int p0, int p1) {
        for (int p = p0; p < p1; p += WAVE) {
            const int mine = p + lane;
            int cj = 0;
            double vj = 0.0;
            if (mine < p1) {
                cj = colidx[mine];
                vj = val[mine];
            }
            const int cnt = min(WAVE, p1 - p);
            for (int g0 = 0; g0 < cnt; g0 += PER_STEP) {
                // slot `sub` of my DPP row takes chunk entry e
                const int e = g0 + half * 16 + sub;
                const int src = e << 2;
                const int gc = __builtin_amdgcn_ds_bpermute(src, cj);
                const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(vj));
                const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(vj));
                const bool on = e < cnt;
                const unsigned co = on ? (unsigned)gc * ldb8 : zero_off;
                const double gv = on ? __hiloint2double(hi, lo) : 0.0;
                const int ng = min(16, cnt - g0); // live slots of the lower half's rows (>= the upper half's)
                unsigned o0, o1, o2, o3, o4, o5, o6, o7;
                double2 b0, b1, b2, b3, b4, b5, b6, b7;
#define SBLAS_LD(O) (*reinterpret_cast<const double2 *>(bt_bytes + (O)))
                SBLAS_DPP_OFF4(0, 1, 2, 3, o0, o1, o2, o3);
                b0 = SBLAS_LD(o0); b1 = SBLAS_LD(o1); b2 = SBLAS_LD(o2); b3 = SBLAS_LD(o3);
                if (ng > 4) {
                    SBLAS_DPP_OFF4(4, 5, 6, 7, o4, o5, o6, o7);
                    b4 = SBLAS_LD(o4); b5 = SBLAS_LD(o5); b6 = SBLAS_LD(o6); b7 = SBLAS_LD(o7);
                    SBLAS_DPP_FMA4x2(0, 1, 2, 3, b0, b1, b2, b3);
                    if (ng > 8) {
                        SBLAS_DPP_OFF4(8, 9, 10, 11, o0, o1, o2, o3);
                        b0 = SBLAS_LD(o0); b1 = SBLAS_LD(o1); b2 = SBLAS_LD(o2); b3 = SBLAS_LD(o3);
                        SBLAS_DPP_FMA4x2(4, 5, 6, 7, b4, b5, b6, b7);
                        if (ng > 12) {
                            SBLAS_DPP_OFF4(12, 13, 14, 15, o4, o5, o6, o7);
                            b4 = SBLAS_LD(o4); b5 = SBLAS_LD(o5); b6 = SBLAS_LD(o6); b7 = SBLAS_LD(o7);
                            SBLAS_DPP_FMA4x2(8, 9, 10, 11, b0, b1, b2, b3);
                            SBLAS_DPP_FMA4x2(12, 13, 14, 15, b4, b5, b6, b7);
                        } else {
                            SBLAS_DPP_FMA4x2(8, 9, 10, 11, b0, b1, b2, b3);
                        }
                    } else {
                        SBLAS_DPP_FMA4x2(4, 5, 6, 7, b4, b5, b6, b7);
                    }
                } else {
                    SBLAS_DPP_FMA4x2(0, 1, 2, 3, b0, b1, b2, b3);
                }
#undef SBLAS_LD
            }
        }
    };
    // Skewed matrices: a row of tens of thousands of entries would keep ONE wave busy while the other fifteen of the
    // workgroup -- and, at the tail of the launch, the whole chip -- wait for it.  Rows of DPP_LONG+ entries are set
    // aside and computed by all sixteen waves together afterwards (each wave a slice of whole 64-entry chunks, the
    // sixteen partial sums added in LDS in wave order), the rule of spmm_direct_rows_kernel.
    __shared__ int long_wave[WIDE_PANEL];
    __shared__ int n_long;
    if (threadIdx.x == 0) n_long = 0;
    __syncthreads();
    int rp0 = 0, rp1 = 0;
    if (mine) {
        rp0 = wave_uniform(rowptr[row]);
        rp1 = wave_uniform(rowptr[row + 1]);
    }
    const bool is_long = mine && rp1 - rp0 >= long_min;
    if (is_long && lane == 0) long_wave[atomicAdd(&n_long, 1)] = wave;
    if (mine && !is_long) sweep(rp0, rp1);
    if (GROUPS == 4) { // the lane groups summed different nonzeros of the same row
        acc0 += __shfl_xor(acc0, 16, WAVE);
        acc1 += __shfl_xor(acc1, 16, WAVE);
    }
    if (GROUPS >= 2) {
        acc0 += __shfl_xor(acc0, 32, WAVE);
        acc1 += __shfl_xor(acc1, 32, WAVE);
    }
    __shared__ int row_mine[WIDE_PANEL];
    if (lane == 0) row_mine[wave] = mine ? 1 : 0;
    if (lane < GLANES) {
        const int cl = 2 * lane;
        ctile[cl][wave] = acc0;
        ctile[cl + 1][wave] = acc1;
    }
    __syncthreads();
    if (n_long > 0) { // (workgroup-uniform; no trips for all but a few panels of a skewed matrix)
        __shared__ double lpart[WIDE_WAVES][TILE_COLS];
        for (int i = 0; i < n_long; ++i) {
            const int lw = long_wave[i];
            const int pa = rowptr[row0 + lw], pb = rowptr[row0 + lw + 1];
            const int slice = ((pb - pa + WIDE_WAVES - 1) / WIDE_WAVES + 63) & ~63; // whole 64-entry chunks per wave
            const int s0 = min(pa + wave * slice, pb), s1 = min(s0 + slice, pb);
            acc0 = acc1 = 0.0;
            sweep(s0, s1);
            if (GROUPS == 4) {
                acc0 += __shfl_xor(acc0, 16, WAVE);
                acc1 += __shfl_xor(acc1, 16, WAVE);
            }
            if (GROUPS >= 2) {
                acc0 += __shfl_xor(acc0, 32, WAVE);
                acc1 += __shfl_xor(acc1, 32, WAVE);
            }
            if (lane < GLANES) {
                lpart[wave][2 * lane] = acc0;
                lpart[wave][2 * lane + 1] = acc1;
            }
            __syncthreads();
            if (threadIdx.x < TILE_COLS) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < WIDE_WAVES; ++w) t += lpart[w][threadIdx.x];
                ctile[threadIdx.x][lw] = t;
            }
            __syncthreads();
        }
    }
    const int nrows = min(WIDE_PANEL, rows - row0);
    const int ncols = min(TILE_COLS, n - col0);
    for (int idx = threadIdx.x; idx < TILE_COLS * WIDE_PANEL; idx += WIDE_WAVES * 64) {
        const int r = idx % WIDE_PANEL, j = idx / WIDE_PANEL;
        if (r < nrows && j < ncols && row_mine[r]) {
            double *dst = C + (int64_t)(col0 + j) * ldc + (row0 + r);
            const double sres = alpha * ctile[j][r];
            *dst = (beta == 0.0) ? sres : fma(beta, *dst, sres);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 2, direct form for rows that share their column pattern (multi-dof FEM matrices: the rows of one mesh node --
// Queen_4147 has three per node -- list the same columns).  The row-per-wave kernel above pulls every Bt row piece
// through the L2 -> CU path once per ROW (grid-structured Queen-like rows at N = 256: 40 GB per call, 1.8 ms at the
// 22 TB/s that path moves); here a wave owns MR = 3 consecutive rows and walks them chunk by chunk (64 entries):
// where the three rows hold the same columns in a chunk, every 16-byte piece of Bt is loaded ONCE and feeds three
// accumulator pairs (one DPP-broadcast value per row); where they differ the chunk falls back to one sweep per row.
// Nothing is assumed: the comparison is made per chunk on the entries themselves.
// ---------------------------------------------------------------------------------------------
constexpr int MR = 3;
constexpr int MERGE_WAVES = 8;               // 24 rows per workgroup, two workgroups per CU
constexpr int MERGE_PANEL = MERGE_WAVES * MR;
template <int NR>
__device__ __forceinline__ void merged_sweeps(int cj, const double (&vj)[NR], int cnt, double (&acc)[NR][2], int sub,
                                              unsigned ldb8, unsigned lb, unsigned zero_off,
                                              const char *__restrict__ bt_bytes)
{
    for (int g0 = 0; g0 < cnt; g0 += 16) {
        const int e = g0 + sub; // slot `sub` of every DPP row takes chunk entry e
        const int src = e << 2;
        const bool on = e < cnt;
        const int gc = __builtin_amdgcn_ds_bpermute(src, cj);
        const unsigned co = on ? (unsigned)gc * ldb8 : zero_off;
        double gvr[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(vj[r]));
            const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(vj[r]));
            gvr[r] = on ? __hiloint2double(hi, lo) : 0.0;
        }
        const int ng = min(16, cnt - g0);
        unsigned o0, o1, o2, o3, o4, o5, o6, o7, o8, o9, o10, o11, o12, o13, o14, o15;
        double2 b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11, b12, b13, b14, b15;
#define SBLAS_LD(O) (*reinterpret_cast<const double2 *>(bt_bytes + (O)))
        // every piece of the sweep in flight before the first FMA
        SBLAS_DPP_OFF4(0, 1, 2, 3, o0, o1, o2, o3);
        b0 = SBLAS_LD(o0); b1 = SBLAS_LD(o1); b2 = SBLAS_LD(o2); b3 = SBLAS_LD(o3);
        if (ng > 4) {
            SBLAS_DPP_OFF4(4, 5, 6, 7, o4, o5, o6, o7);
            b4 = SBLAS_LD(o4); b5 = SBLAS_LD(o5); b6 = SBLAS_LD(o6); b7 = SBLAS_LD(o7);
        }
        if (ng > 8) {
            SBLAS_DPP_OFF4(8, 9, 10, 11, o8, o9, o10, o11);
            b8 = SBLAS_LD(o8); b9 = SBLAS_LD(o9); b10 = SBLAS_LD(o10); b11 = SBLAS_LD(o11);
        }
        if (ng > 12) {
            SBLAS_DPP_OFF4(12, 13, 14, 15, o12, o13, o14, o15);
            b12 = SBLAS_LD(o12); b13 = SBLAS_LD(o13); b14 = SBLAS_LD(o14); b15 = SBLAS_LD(o15);
        }
#undef SBLAS_LD
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const double gv = gvr[r];
            double &acc0 = acc[r][0], &acc1 = acc[r][1];
            SBLAS_DPP_FMA4x2(0, 1, 2, 3, b0, b1, b2, b3);
            if (ng > 4) { SBLAS_DPP_FMA4x2(4, 5, 6, 7, b4, b5, b6, b7); }
            if (ng > 8) { SBLAS_DPP_FMA4x2(8, 9, 10, 11, b8, b9, b10, b11); }
            if (ng > 12) { SBLAS_DPP_FMA4x2(12, 13, 14, 15, b12, b13, b14, b15); }
        }
    }
}

__global__ __launch_bounds__(MERGE_WAVES * 64) void spmm_direct_merge_kernel(
    int rows, int cols, int npanels, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, const int *__restrict__ tail, const int *__restrict__ cls,
    int info_panel_rows, int interleave, int epoch)
{
    constexpr int TILE_COLS = 128;
    extern __shared__ __attribute__((aligned(16))) double merge_smem[];
    double(*ctile)[MERGE_PANEL + 1] = reinterpret_cast<double(*)[MERGE_PANEL + 1]>(merge_smem); // [TILE_COLS][MERGE_PANEL + 1]
    __shared__ int row_mine[MERGE_PANEL];
    if (cls != nullptr && (nothing_direct(tail, epoch) || tail[TAIL_MERGE_EPOCH] != epoch)) return; // the row-per-wave kernel's call
    const int lane = threadIdx.x & 63;
    const int wave = wave_uniform(threadIdx.x >> 6);
    if (interleave < 0) { // wide bands: neighbouring panels on different XCDs (see spmm_direct_dpp_kernel)
        interleave = 0;
        if (cls != nullptr) interleave = (long long)tail[TAIL_BAND] * (TILE_COLS * 8) > (16ll << 20);
    }
    const int row0 = (interleave ? (int)blockIdx.x : xcd_contiguous_panel(blockIdx.x, npanels)) * MERGE_PANEL;
    const int col0 = blockIdx.y * TILE_COLS;
    const int sub = lane & 15;
    const unsigned ldb8 = (unsigned)ldbt * 8u;
    const unsigned lb = (unsigned)(col0 * 8) + (unsigned)lane * 16u; // this lane's 2 columns
    const unsigned zero_off = (unsigned)cols * ldb8;
    const char *__restrict__ bt_bytes = reinterpret_cast<const char *>(Bt);

    int p0[MR], len[MR];
    bool mine[MR];
    bool all_mine = true;
    // A method-2 row block starts anywhere, so groups of three pattern-sharing rows need not start at a multiple of
    // three: the classifier found where they do (class word), waves 1.. take the groups from there and wave 0 the
    // rows left over at the two ends of the panel (a group of its own when the phase is 0).
    static_assert(MERGE_PANEL % MR == 0 && MERGE_PANEL == MERGE_WAVES * MR && MR == 3, "row groups of the merging kernel");
    int phase = 0;
    if (cls != nullptr) phase = wave_uniform((cls[min(row0, rows - 1) / info_panel_rows] >> PANEL_PHASE_SHIFT) & 3) % MR;
    int lr[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r)
        lr[r] = wave > 0 ? phase + MR * (wave - 1) + r : (r < phase ? r : MERGE_PANEL - MR + r);
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        const int row = row0 + lr[r];
        mine[r] = row < rows;
        if (cls && mine[r]) {
            mine[r] = owns_direct(tail, cls, row / info_panel_rows);
            if (mine[r] && lane == 0 && blockIdx.y == 0 && row % info_panel_rows == 0) atomicAdd(&g_panel_stats[1], 1ull);
        }
        p0[r] = len[r] = 0;
        if (mine[r]) {
            p0[r] = wave_uniform(rowptr[row]);
            len[r] = wave_uniform(rowptr[row + 1]) - p0[r];
        }
        all_mine = all_mine && mine[r];
    }
    double acc[MR][2];
#pragma unroll
    for (int r = 0; r < MR; ++r) acc[r][0] = acc[r][1] = 0.0;
    const int maxlen = max(len[0], max(len[1], len[2]));
    for (int q = 0; q < maxlen; q += WAVE) {
        int cj[MR], cnt[MR];
        double vj[MR];
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            cnt[r] = min(WAVE, max(len[r] - q, 0));
            cj[r] = 0;
            vj[r] = 0.0;
            if (lane < cnt[r]) {
                cj[r] = colidx[p0[r] + q + lane];
                vj[r] = val[p0[r] + q + lane];
            }
        }
        const bool same = all_mine && cnt[1] == cnt[0] && cnt[2] == cnt[0] &&
                          __builtin_amdgcn_ballot_w64(lane < cnt[0] && (cj[1] != cj[0] || cj[2] != cj[0])) == 0ull;
        if (same) {
            merged_sweeps<MR>(cj[0], vj, cnt[0], acc, sub, ldb8, lb, zero_off, bt_bytes);
        } else {
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                if (cnt[r] > 0) { // (wave-uniform)
                    const double v1[1] = {vj[r]};
                    double a1[1][2] = {{acc[r][0], acc[r][1]}};
                    merged_sweeps<1>(cj[r], v1, cnt[r], a1, sub, ldb8, lb, zero_off, bt_bytes);
                    acc[r][0] = a1[0][0];
                    acc[r][1] = a1[0][1];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        if (lane == 0) row_mine[lr[r]] = mine[r] ? 1 : 0;
        ctile[2 * lane][lr[r]] = acc[r][0];
        ctile[2 * lane + 1][lr[r]] = acc[r][1];
    }
    __syncthreads();
    const int nrows = min(MERGE_PANEL, rows - row0);
    const int ncols = min(TILE_COLS, n - col0);
    for (int idx = threadIdx.x; idx < TILE_COLS * MERGE_PANEL; idx += MERGE_WAVES * 64) {
        const int r = idx % MERGE_PANEL, j = idx / MERGE_PANEL;
        if (r < nrows && j < ncols && row_mine[r]) {
            double *dst = C + (int64_t)(col0 + j) * ldc + (row0 + r);
            const double sres = alpha * ctile[j][r];
            *dst = (beta == 0.0) ? sres : fma(beta, *dst, sres);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 2, narrow form (ldbt = G in {8,16,32}, n <= G): a group of G lanes owns a row, so a wave
// works on 64/G rows at once and every lane fetches its row's (col, val) itself (the G lanes of a
// group read the same address, which the memory pipeline serves as one request).
// ---------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void spmm_rowpanel_narrow_kernel(int rows, const int *__restrict__ rowptr,
                                                                  const int *__restrict__ colidx,
                                                                  const double *__restrict__ val,
                                                                  const double *__restrict__ Bt, int n,
                                                                  double alpha, double beta,
                                                                  double *__restrict__ C, int64_t ldc,
                                                                  const int *__restrict__ tail,
                                                                  const int *__restrict__ cls, int info_panel_rows,
                                                                  int epoch)
{
    constexpr int GROUPS = 256 / G;
    constexpr int RPG = PANEL_ROWS / GROUPS; // rows per group
    static_assert(RPG >= 1, "panel too small for this group width");
    __shared__ double ctile[G][PANEL_ROWS + 1];
    __shared__ int row_mine[PANEL_ROWS];
    if (cls != nullptr && nothing_direct(tail, epoch)) return; // every panel went to the LDS-tiled kernel
    const int l = threadIdx.x % G, grp = threadIdx.x / G;
    const int row0 = blockIdx.x * PANEL_ROWS;
    for (int rr = 0; rr < RPG; ++rr) {
        const int r = grp * RPG + rr;
        const int row = row0 + r;
        double acc = 0.0;
        bool mine = row < rows;
        if (cls && mine) {
            mine = owns_direct(tail, cls, row / info_panel_rows);
            if (mine && l == 0 && row % info_panel_rows == 0) atomicAdd(&g_panel_stats[1], 1ull);
        }
        if (mine) {
            const int p0 = rowptr[row], p1 = rowptr[row + 1];
            int p = p0;
            for (; p + 4 <= p1; p += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc = fma(val[p + u], Bt[(int64_t)colidx[p + u] * G + l], acc);
            }
            for (; p < p1; ++p) acc = fma(val[p], Bt[(int64_t)colidx[p] * G + l], acc);
        }
        ctile[l][r] = acc;
        if (l == 0) row_mine[r] = mine ? 1 : 0;
    }
    __syncthreads();
    const int nrows = min(PANEL_ROWS, rows - row0);
    for (int idx = threadIdx.x; idx < G * PANEL_ROWS; idx += 256) {
        const int r = idx % PANEL_ROWS, j = idx / PANEL_ROWS;
        if (r < nrows && j < n && row_mine[r]) {
            double *dst = C + (int64_t)j * ldc + (row0 + r);
            const double s = alpha * ctile[j][r];
            *dst = (beta == 0.0) ? s : fma(beta, *dst, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 2, direct form for SHORT rows: four matrix rows per wave, one per DPP row (the layout of the sixth windowed
// generation, reading Bt from L2 instead of LDS).  With a row per wave a 5-nonzero row occupies a 16-slot sweep, a
// 64-lane chunk load, a reduction and a C write of its own: 0.8 ms for a million such rows at N = 64 (17 % of the
// HBM time).  Here lane k of DPP row q holds entry k of row 4w+q, step k serves the k-th nonzero of four rows at once,
// a lane accumulates four columns of its row (two 16-byte loads per step), and a workgroup writes 64 rows of C.
// Used for the direct panels when the matrix averages fewer than 56 (64 staged columns) / 32 (128+) nonzeros per row, and
// from 128 staged columns on wherever the classifier's vote prefers it to a row per wave (classify_panel).
// ---------------------------------------------------------------------------------------------
constexpr int ROWS_LONG = 512; // entries from which a row is computed by the whole workgroup
template <int WV> // waves per workgroup (4 WV rows)
__global__ __launch_bounds__(WV * 64) void spmm_direct_rows_kernel(
    int rows, int cols, int npanels, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, const int *__restrict__ tail, const int *__restrict__ cls,
    int info_panel_rows, int interleave, int epoch, int voted)
{
    constexpr int ROWS_PANEL = 4 * WV;
    __shared__ double ctile[64][ROWS_PANEL + 1];
    __shared__ int row_mine[ROWS_PANEL];
    if (cls != nullptr && nothing_direct(tail, epoch)) return;
    if (voted && tail[TAIL_ROWS_EPOCH] != epoch) return; // (128+ staged columns: the vote gave the call to another direct kernel)
    const int lane = threadIdx.x & 63;
    const int wave = wave_uniform(threadIdx.x >> 6);
    const int k = lane & 15, q = lane >> 4;
    const int col0 = blockIdx.y * 64;
    if (col0 >= n) return; // a padding tile of ldbt (uniform over the workgroup)
    if (interleave < 0) {
        interleave = 0;
        if (cls != nullptr) {
            const int band = tail[TAIL_BAND];
            interleave = (long long)band * 512 > (16ll << 20);
        }
    }
    const int row0 = (interleave ? (int)blockIdx.x : xcd_contiguous_panel(blockIdx.x, npanels)) * ROWS_PANEL;
    const int rr = wave * 4 + q;
    const int row = row0 + rr;
    bool mine = row < rows;
    if (cls && mine) {
        mine = owns_direct(tail, cls, row / info_panel_rows);
        if (mine && k == 0 && blockIdx.y == 0 && row % info_panel_rows == 0) atomicAdd(&g_panel_stats[1], 1ull);
    }
    const unsigned ldb8 = (unsigned)ldbt * 8u;
    const unsigned lb = (unsigned)(col0 * 8) + (unsigned)k * 16u; // columns 2k, 2k+1 (and 32+2k, 33+2k at +256 bytes)
    const unsigned zero_off = (unsigned)cols * ldb8;
    const char *__restrict__ bt_bytes = reinterpret_cast<const char *>(Bt);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int p = 0, pend = 0;
    if (mine) {
        p = rowptr[row];
        pend = rowptr[row + 1];
    }
    // Skewed matrices (the kernel is picked by the AVERAGE row length): a row of thousands of entries would walk
    // them sixteen at a time on one DPP row while the other 63 rows of the panel wait.  Such rows are set aside here
    // and computed by all sixteen waves together after the sweep (each wave a slice of the row, sums merged in LDS).
    __shared__ int long_rr[ROWS_PANEL];
    __shared__ int n_long;
    __shared__ double lpart[WV][64];
    if (threadIdx.x == 0) n_long = 0;
    __syncthreads();
    if (mine && pend - p > ROWS_LONG) {
        if (k == 0) long_rr[atomicAdd(&n_long, 1)] = rr;
        p = pend = 0;
    }
    for (int base = p; __builtin_amdgcn_ballot_w64(base < pend) != 0ull; base += 16) {
        const int idx = base + k;
        const bool valid = idx < pend;
        int c = 0;
        double v = 0.0;
        if (valid) {
            c = colidx[idx];
            v = val[idx];
        }
        const unsigned co = valid ? (unsigned)c * ldb8 : zero_off;
        const double gv = valid ? v : 0.0;
        const int left = min(16, max(pend - base, 0)); // the same in all 16 lanes of a DPP row
        const int mx = max(max(__builtin_amdgcn_readlane(left, 0), __builtin_amdgcn_readlane(left, 16)),
                           max(__builtin_amdgcn_readlane(left, 32), __builtin_amdgcn_readlane(left, 48)));
#define SBLAS_LD(O) (*reinterpret_cast<const double2 *>(bt_bytes + (O)))
#define SBLAS_ROWS_BLOCK(K0, K1, K2, K3)                                                                              \
    {                                                                                                                \
        unsigned o0, o1, o2, o3;                                                                                     \
        SBLAS_DPP_OFF4(K0, K1, K2, K3, o0, o1, o2, o3);                                                              \
        const double2 b0 = SBLAS_LD(o0), b1 = SBLAS_LD(o1), b2 = SBLAS_LD(o2), b3 = SBLAS_LD(o3);                    \
        const double2 d0 = SBLAS_LD(o0 + 256u), d1 = SBLAS_LD(o1 + 256u), d2 = SBLAS_LD(o2 + 256u),                  \
                      d3 = SBLAS_LD(o3 + 256u);                                                                      \
        {                                                                                                            \
            double &acc0 = a0, &acc1 = a1;                                                                           \
            SBLAS_DPP_FMA4x2(K0, K1, K2, K3, b0, b1, b2, b3);                                                        \
        }                                                                                                            \
        {                                                                                                            \
            double &acc0 = a2, &acc1 = a3;                                                                           \
            SBLAS_DPP_FMA4x2(K0, K1, K2, K3, d0, d1, d2, d3);                                                        \
        }                                                                                                            \
    }
        SBLAS_ROWS_BLOCK(0, 1, 2, 3)
        if (mx > 4) SBLAS_ROWS_BLOCK(4, 5, 6, 7)
        if (mx > 8) SBLAS_ROWS_BLOCK(8, 9, 10, 11)
        if (mx > 12) SBLAS_ROWS_BLOCK(12, 13, 14, 15)
#undef SBLAS_ROWS_BLOCK
#undef SBLAS_LD
    }
    if (k == 0) row_mine[rr] = mine ? 1 : 0;
    ctile[2 * k][rr] = a0;
    ctile[2 * k + 1][rr] = a1;
    ctile[32 + 2 * k][rr] = a2;
    ctile[33 + 2 * k][rr] = a3;
    __syncthreads();
    for (int i = 0; i < n_long; ++i) { // (workgroup-uniform; no trips for all but a few panels)
        const int lrr = long_rr[i];
        const int pa = rowptr[row0 + lrr], pb = rowptr[row0 + lrr + 1];
        const int slice = ((pb - pa + WV - 1) / WV + 63) & ~63; // whole 64-entry chunks per wave
        const int s0 = min(pa + wave * slice, pb), s1 = min(s0 + slice, pb);
        lpart[wave][lane] = row_direct(colidx, val, Bt, (unsigned)ldbt, (unsigned)(col0 + lane), lane, s0, s1);
        __syncthreads();
        if (threadIdx.x < 64) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < WV; ++w) t += lpart[w][threadIdx.x];
            ctile[threadIdx.x][lrr] = t;
        }
        __syncthreads();
    }
    const int nrows = min(ROWS_PANEL, rows - row0);
    const int ncols = min(64, n - col0);
    for (int idx = threadIdx.x; idx < 64 * ROWS_PANEL; idx += WV * 64) {
        const int r = idx % ROWS_PANEL, j = idx / ROWS_PANEL;
        if (r < nrows && j < ncols && row_mine[r]) {
            double *dst = C + (int64_t)(col0 + j) * ldc + (row0 + r);
            const double sres = alpha * ctile[j][r];
            *dst = (beta == 0.0) ? sres : fma(beta, *dst, sres);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SpMM with at most 8 columns (ldbt = 8) and LONG rows (256+ on average): "SpMV with eight right-hand sides".  The lane-group kernel
// above walks a row's nonzeros serially in an 8-lane group (0.33 ms on the bench matrix whatever N <= 8 is -- method
// 1 on eight GPUs hands every GPU 8 of 64 columns).  Here a wave owns a row, its 64 lanes stride through the
// nonzeros (coalesced col_idx / val streams, four slices in flight), every lane reads the 64-byte Bt row of its
// nonzero and keeps eight partial sums; the eight sums are folded across the wave by a halving exchange (4 + 2 + 1
// shuffles, then three more: ten instead of 48) and lanes 0, 8, .., 56 write columns 0..7.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spmm_rows8_kernel(int rows, int cols, const int *__restrict__ rowptr,
                                                        const int *__restrict__ colidx,
                                                        const double *__restrict__ val,
                                                        const double *__restrict__ Bt, int n, double alpha,
                                                        double beta, double *__restrict__ C, int64_t ldc,
                                                        const int *__restrict__ tail, const int *__restrict__ cls,
                                                        int info_panel_rows, int epoch)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave_uniform(threadIdx.x >> 6);
    if (row >= rows) return;
    if (cls != nullptr) { // rows of panels the LDS-tiled kernel owns are skipped (a row per wave: wave-uniform)
        if (nothing_direct(tail, epoch) || !owns_direct(tail, cls, row / info_panel_rows)) return;
        if (lane == 0 && row % info_panel_rows == 0) atomicAdd(&g_panel_stats[1], 1ull);
    }
    const int p0 = wave_uniform(rowptr[row]), p1 = wave_uniform(rowptr[row + 1]);
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.0;
    for (int p = p0 + lane; p < p1; p += 4 * WAVE) { // (a lane past the end reads the all-zero row Bt[cols])
        int c[4];
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = p + u * WAVE;
            c[u] = q < p1 ? colidx[q] : cols;
            v[u] = q < p1 ? val[q] : 0.0;
        }
        double2 b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double2 *__restrict__ br = reinterpret_cast<const double2 *>(Bt + (int64_t)c[u] * 8);
#pragma unroll
            for (int h = 0; h < 4; ++h) b[u][h] = br[h];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                a[2 * h] = fma(v[u], b[u][h].x, a[2 * h]);
                a[2 * h + 1] = fma(v[u], b[u][h].y, a[2 * h + 1]);
            }
    }
    // halving exchange: after the step with mask m a lane keeps the half of its sums selected by (lane & m)
    double b4[4], b2[2], s;
    {
        const bool hi = lane & 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) b4[j] = (hi ? a[j + 4] : a[j]) + __shfl_xor(hi ? a[j] : a[j + 4], 32, WAVE);
    }
    {
        const bool hi = lane & 16;
#pragma unroll
        for (int j = 0; j < 2; ++j) b2[j] = (hi ? b4[j + 2] : b4[j]) + __shfl_xor(hi ? b4[j] : b4[j + 2], 16, WAVE);
    }
    {
        const bool hi = lane & 8;
        s = (hi ? b2[1] : b2[0]) + __shfl_xor(hi ? b2[0] : b2[1], 8, WAVE);
    }
    s += __shfl_xor(s, 4, WAVE);
    s += __shfl_xor(s, 2, WAVE);
    s += __shfl_xor(s, 1, WAVE);
    const int j = lane >> 3; // column of this lane's sum: (lane & 32 ? 4 : 0) + (lane & 16 ? 2 : 0) + (lane & 8 ? 1 : 0)
    if ((lane & 7) == 0 && j < n) {
        double *dst = C + (int64_t)j * ldc + row;
        const double r = alpha * s;
        *dst = (beta == 0.0) ? r : fma(beta, *dst, r);
    }
}

// ---------------------------------------------------------------------------------------------
// y = beta*y + alpha*x  (kernel.h:27-38), two doubles per lane per step, grid-stride.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void axpby_kernel(int64_t n, double alpha, const double *__restrict__ x,
                                                   double beta, double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool aligned = ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
    if (aligned) {
        const int64_t n2 = n >> 1;
        const double2 *x2 = reinterpret_cast<const double2 *>(x);
        double2 *y2 = reinterpret_cast<double2 *>(y);
        for (int64_t i = tid; i < n2; i += stride) {
            double2 a = x2[i], b = y2[i];
            b.x = b.x * beta + a.x * alpha;
            b.y = b.y * beta + a.y * alpha;
            y2[i] = b;
        }
        if ((n & 1) && tid == 0) y[n - 1] = y[n - 1] * beta + x[n - 1] * alpha;
    } else {
        for (int64_t i = tid; i < n; i += stride) y[i] = y[i] * beta + x[i] * alpha;
    }
}

// C = beta * C on a rows x n column-major block (a matrix without nonzeros: A*B = 0).  beta = 0 stores zeros without
// reading C (BLAS semantics: NaNs in C do not propagate).
__global__ __launch_bounds__(256) void scale_kernel(int64_t rows, int64_t n, double beta, double *__restrict__ C,
                                                   int64_t ldc)
{
    const int64_t total = rows * n, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        double *dst = C + (i / rows) * ldc + (i % rows);
        *dst = (beta == 0.0) ? 0.0 : beta * *dst;
    }
}

// Method-2 merge without an all-reduce.  The partial results of the row-block scheme are disjoint
// except for the rows a block boundary cuts, so every GPU only needs the OTHER GPUs' blocks (packed, m_q x N,
// leading dimension m_q) and one pass that scatters them into place and applies alpha / beta:
//   C[r, j] = beta * C[r, j] + alpha * sum over the blocks q that contain row r of src_q[r - start_q, j]
// (a boundary row gets two terms, a row longer than nnz/g more).  Replaces the M x N zero fill, the all-reduce of
// the full M x N buffer and the axpby pass of spmm.h:222-283 / spmv.h:60-138.
struct RowBlocks {
    const double *src[MAX_REPLICAS];
    long long start[MAX_REPLICAS];
    long long nrows[MAX_REPLICAS];
};
__global__ __launch_bounds__(256) void merge_rowblocks_kernel(long long M, long long N, int g, RowBlocks b, double alpha,
                                                             double beta, double *__restrict__ C, long long ldc)
{
    const long long total = M * N, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long r = i % M, j = i / M;
        double s = 0.0;
        for (int q = 0; q < g; ++q) {
            const long long rel = r - b.start[q];
            if (rel >= 0 && rel < b.nrows[q]) s += b.src[q][j * b.nrows[q] + rel];
        }
        double *dst = C + j * ldc + r;
        const double res = alpha * s;
        *dst = (beta == 0.0) ? res : fma(beta, *dst, res);
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline unsigned capped_grid(int64_t work_items, int per_block)
{
    int64_t b = (work_items + per_block - 1) / per_block;
    const int64_t cap = 256 * 8; // CUs x resident blocks (guide: cap memory-bound grids, stride the rest)
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}


// Opt-in check of the CONTENTS of a CSR structure (sblas_hip_debug_validate_csr_i32, SBLAS_VALIDATE=1): the compute
// kernels trust them, as the vendor libraries do -- a column index outside [0, cols) is an out-of-bounds read of Bt.
// flag[0] |= 1: a row pointer runs backwards or past nnz; 2: rowptr[0] != 0 or rowptr[rows] != nnz; 4: a column index
// outside [0, cols).
__global__ __launch_bounds__(256) void validate_csr_kernel(int64_t rows, int64_t cols, int64_t nnz, const int *__restrict__ rowptr,
                                                           const int *__restrict__ colidx, int *__restrict__ flag)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int bad = 0;
    for (int64_t r = tid; r < rows; r += stride) {
        const int a = rowptr[r], b = rowptr[r + 1];
        if (a > b || a < 0 || (int64_t)b > nnz) bad |= 1;
    }
    if (tid == 0 && (rowptr[0] != 0 || (int64_t)rowptr[rows] != nnz)) bad |= 2;
    for (int64_t k = tid; k < nnz; k += stride) {
        const int c = colidx[k];
        if (c < 0 || (int64_t)c >= cols) bad |= 4;
    }
    if (bad) atomicOr(flag, bad);
}
hipError_t validate_csr(hipStream_t s, int64_t rows, int64_t cols, int64_t nnz, const int *rowptr, const int *colidx, int *bad)
{
    int *flag = nullptr;
    hipError_t e = hipMalloc(&flag, sizeof(int));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(flag, 0, sizeof(int), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(validate_csr_kernel, dim3(capped_grid(std::max<int64_t>(rows, nnz), 256)), dim3(256), 0, s, rows, cols, nnz,
                           rowptr, colidx, flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(bad, flag, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(flag);
    return e;
}

// compute units of the current device (queried once per device)
static int compute_units()
{
    static std::atomic<int> cached[16]; // (zero-initialised; any thread may fill a slot, all write the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// Diagnostics: two events around the dominant stage-2 kernel of the most recent launch on a device (see
// sblas_hip_debug_spmm_kernel_events).  NOT thread-safe (a measuring harness has one launching thread); likewise
// sblas_hip_debug_reload_env rewrites the switches in place and must not run beside launches of other threads.
namespace {
bool g_kernel_events = false;
struct KernelEvents {
    hipEvent_t a = nullptr, b = nullptr;
    bool recorded = false;
} g_kev[16];
KernelEvents *kernel_events_slot()
{
    int dev = 0;
    if (!g_kernel_events || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    KernelEvents &e = g_kev[dev];
    if (!e.a && (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess)) return nullptr;
    return &e;
}
} // namespace
void kernel_events_enable(bool on) { g_kernel_events = on; }
hipError_t kernel_events_last_ms(float *ms)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || !g_kev[dev].recorded) return hipErrorNotReady;
    hipError_t e = hipEventSynchronize(g_kev[dev].b);
    if (e != hipSuccess) return e;
    return hipEventElapsedTime(ms, g_kev[dev].a, g_kev[dev].b);
}

// ---------------------------------------------------------------------------------------------
// Experiment / test switches.  The environment is read ONCE (first launch) into this struct; tests that change a
// switch inside one process call sblas_hip_debug_reload_env() afterwards.  Nothing here changes results.
// ---------------------------------------------------------------------------------------------
static Options g_options;
static std::atomic<bool> g_options_loaded{false};
static std::mutex g_options_mu;
static void options_parse(Options &o)
{
    o = Options{};
    const char *e;
    if ((e = getenv("SBLAS_SPMM_VARIANT")) && *e) {
        if (!strcmp(e, "dpp")) o.spmm_variant = SPMM_VARIANT_DIRECT_DPP;
        else if (!strcmp(e, "rows")) o.spmm_variant = SPMM_VARIANT_DIRECT_ROWS;
        else if (!strcmp(e, "lanes")) o.spmm_variant = SPMM_VARIANT_LANES;
        else if (!strcmp(e, "merge")) o.spmm_variant = SPMM_VARIANT_DIRECT_MERGE;
        else if (!strcmp(e, "mfma")) o.spmm_variant = SPMM_VARIANT_MFMA;
        else if (!strcmp(e, "nomfma")) o.spmm_variant = SPMM_VARIANT_NO_MFMA;
    }
    if ((e = getenv("SBLAS_SPMV_VARIANT")) && *e && strcmp(e, "auto")) {
        strncpy(o.spmv_variant, e, sizeof o.spmv_variant - 1);
    }
    if ((e = getenv("SBLAS_SPMM_MIN_LDBT")) && *e) o.tier16 = o.tier32 = atoi(e) < 64;
    if ((e = getenv("SBLAS_SPMM_MAX_BT_BYTES")) && *e) {
        const unsigned long long v = strtoull(e, nullptr, 10);
        if (v >= 4096 && v < 0xffffffffull) o.max_bt_bytes = v;
    }
    if ((e = getenv("SBLAS_DIRECT_LDS")) && *e) o.direct_lds = atoi(e);
    if ((e = getenv("SBLAS_DIRECT_MERGE")) && *e) o.direct_merge = atoi(e);
    if ((e = getenv("SBLAS_STAGE_RANGE")) && *e) o.stage_range = atoi(e);
    if ((e = getenv("SBLAS_DIRECT_MAP")) && *e) o.direct_map = !strcmp(e, "interleave") ? 1 : !strcmp(e, "contiguous") ? 0 : -1;
    if ((e = getenv("SBLAS_ROWS8_MIN_AVG")) && *e) o.rows8_min_avg = atof(e);
    if ((e = getenv("SBLAS_WINDOW_DENSITY")) && *e) o.window_density = (float)atof(e);
    if ((e = getenv("SBLAS_SPMM_PANEL_ROWS")) && *e) { /* "<rows>" or "<rows>,<groups>" */
        o.panel_rows = atoi(e);
        o.panel_groups = strchr(e, ',') ? atoi(strchr(e, ',') + 1) : 0;
    }
    if ((e = getenv("SBLAS_MFMA_MIN_FILL")) && *e) o.mfma_min_fill = (float)atof(e);
    if ((e = getenv("SBLAS_VALIDATE")) && *e) o.validate = atoi(e) != 0;
    if ((e = getenv("SBLAS_TUNE")) && *e) { /* "a,b,c,d" (or "a:b:c:d"): free integers for kernel experiments */
        char buf[96];
        strncpy(buf, e, sizeof buf - 1);
        buf[sizeof buf - 1] = 0;
        for (char *c = buf; *c; ++c)
            if (*c == ':') *c = ',';
        sscanf(buf, "%d,%d,%d,%d", &o.tune[0], &o.tune[1], &o.tune[2], &o.tune[3]);
    }
}
const Options &options()
{
    if (!g_options_loaded.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(g_options_mu);
        if (!g_options_loaded.load(std::memory_order_relaxed)) {
            options_parse(g_options);
            g_options_loaded.store(true, std::memory_order_release);
        }
    }
    return g_options;
}
void options_reload()
{
    std::lock_guard<std::mutex> lock(g_options_mu);
    options_parse(g_options);
    g_options_loaded.store(true, std::memory_order_release);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is per device: raise it once per (kernel, device), not per launch
void raise_dynamic_lds(const void *fn, size_t bytes)
{
    struct Seen { const void *fn; int dev; size_t bytes; };
    static std::mutex mu;
    static std::vector<Seen> seen;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    for (Seen &q : seen)
        if (q.fn == fn && q.dev == dev) {
            if (q.bytes >= bytes) return;
            q.bytes = bytes;
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            return;
        }
    seen.push_back({fn, dev, bytes});
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// Sixth generation, one workgroup per CU at a time: the groups per wave (2 or 3) and the panel height (a multiple of
// the rows of a wave) that minimise rounds x (height + per-tile fixed cost).
static void gen6_plan(int rows, int &info_rows, int &gen6_g, int gmax = W6_GMAX)
{
    const int ncu = compute_units();
    int best = 128;
    long best_cost = -1;
    gen6_g = 2;
    for (int g = 2; g <= gmax; ++g)
        for (int r = 12 * 4 * g; r >= 4 * 4 * g; r -= 4 * g) {
            const long panels = (rows + r - 1) / r;
            // measured on the bench matrix: three groups per wave cost ~15 % more per row
            const long cost = ((panels + ncu - 1) / ncu) * (long)(r + 40) * (g == 3 ? 23 : 20);
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best = r;
                gen6_g = g;
            }
        }
    info_rows = best;
    const Options &opt = options();
    if (opt.panel_rows > 0) { /* experiments */
        const int r = opt.panel_rows, g = opt.panel_groups ? opt.panel_groups : (r % 12 == 0 && r > 128 ? 3 : 2);
        if (g >= 2 && g <= gmax && r >= SPMM_MIN_PANEL_ROWS && r <= 48 * g && r % (4 * g) == 0) { // (workspace: a verdict per 32 rows)
            info_rows = r;
            gen6_g = g;
        }
    }
}

// Panel height of the narrow LDS-tiled kernel: twelve consumer waves of G groups of four rows; G is capped by the
// registers NC accumulators per group take (8 columns: 3, 16 and 32: 1).
static void lanes_plan(int rows, int ldbt, int &info_rows, int &groups)
{
    const int ncu = compute_units();
    const int gmax = ldbt <= 8 ? 3 : ldbt <= 16 ? (options().tune[0] == 2 ? 2 : 3) : options().tune[3] != 1 ? 2 : 1;
    long best_cost = -1;
    info_rows = 48;
    groups = 1;
    // 8 columns: one group of rows per wave, two workgroups per CU (see spmm_lanes_kernel); SBLAS_SPMM_PANEL_ROWS overrides
    for (int g = 1; g <= gmax && ldbt > 8; ++g)
        for (int r = 12 * 4 * g; r >= 4 * 4 * g; r -= 4 * g) {
            if (r < SPMM_MIN_PANEL_ROWS) continue;
            const long panels = (rows + r - 1) / r;
            const long cost = ((panels + ncu - 1) / ncu) * (long)(r + 40);
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                info_rows = r;
                groups = g;
            }
        }
    const Options &opt = options();
    if (opt.panel_rows > 0) { /* experiments */
        const int r = opt.panel_rows, g = opt.panel_groups ? opt.panel_groups : 1;
        if (g >= 1 && g <= gmax && r >= SPMM_MIN_PANEL_ROWS && r <= 48 * g && r % (4 * g) == 0) {
            info_rows = r;
            groups = g;
        }
    }
}

// the panel height the classifier and the stage-2 kernels of a call agree on
static void panel_plan(int rows, int64_t ldbt, int &info_rows, int &groups)
{
    if (ldbt < 64) lanes_plan(rows, (int)ldbt, info_rows, groups);
    // (128+ staged columns: two column halves per workgroup, whose accumulators leave room for two groups per wave)
    // (... and so does SBLAS_SPMM_VARIANT=mfma at 64 columns: the matrix-core kernel takes panels of up to 16 rows x
    //  MFMA_MAX_WAVES, and three groups per wave would give 132- / 144-row panels it silently leaves alone: ADVICE r2)
    else gen6_plan(rows, info_rows, groups,
                   ((ldbt >= 128 && options().tune[1] != 1) || options().spmm_variant == SPMM_VARIANT_MFMA) ? 2 : W6_GMAX);
}

static std::atomic<int> g_epoch{1}; // tags one call's classifier verdicts and one staging pass (see classify_panel)

// The workspace behind the staging copy (kernels.h): header ints, one span per panel, one class per panel.
struct Tail {
    int *hdr;
    int2 *parts; // TAIL_PARTS (min, max) pairs of the column-range pass
    int2 *info;
    int *cls;
};
static Tail tail_at(int *hdr, int rows);
static Tail tail_of(const double *Bt, int64_t cols, int64_t ldbt, int rows)
{
    return tail_at(reinterpret_cast<int *>(const_cast<double *>(Bt) + (size_t)(cols + 1) * (size_t)ldbt), rows);
}
static Tail tail_at(int *hdr, int rows)
{
    Tail t;
    t.hdr = hdr;
    const size_t panels = ((size_t)(rows > 0 ? rows : 0) + SPMM_MIN_PANEL_ROWS - 1) / SPMM_MIN_PANEL_ROWS;
    t.parts = reinterpret_cast<int2 *>(t.hdr + TAIL_HDR);
    t.info = t.parts + TAIL_PARTS;
    t.cls = reinterpret_cast<int *>(t.info + panels);
    return t;
}
size_t workspace_tail_bytes(int64_t rows)
{
    const size_t panels = ((size_t)(rows > 0 ? rows : 0) + SPMM_MIN_PANEL_ROWS - 1) / SPMM_MIN_PANEL_ROWS;
    return (TAIL_HDR * sizeof(int) + TAIL_PARTS * sizeof(int2) + panels * (sizeof(int2) + sizeof(int)) + 31) / 16 * 16; // whole 16-byte units
}
// block fill from which a panel goes to the matrix cores (fp64 MFMA and fp64 vector FMA have the same peak on gfx950,
// so the zero fill of a block is paid in full): measured break-even against the vector kernels, tools/spmm_shapes.py blocks:ROWS:FILL
// Nonzeros a panel must hold per column of its span to take the LDS-tiled kernel: every row of a B tile it loads is then
// used that many times on average.  0.42 per 16 rows = 2.5 uses for a 96-row panel -- measured on banded rows, 1 M
// rows, N = 64 (tools/spmm_shapes.py banded:ROWS:PERROW:HALFBAND with SBLAS_WINDOW_DENSITY): at 1.4 uses the direct
// kernel wins by 36 %, at 2.1-2.4 the two are within 5 %, from 2.75 on the LDS-tiled kernel wins (6 % ... 70 % at 6 uses,
// where round 1's bar stood).
static float window_min_density(int panel_rows) { return options().window_density * (float)panel_rows / 16.0f; }
// Nonzeros per row a panel must average to take the narrow LDS-tiled kernel (0 at 64+ staged columns).  Measured on banded rows
// (tools/spmm_shapes.py banded:ROWS:PERROW:HALFBAND, 1 M rows) against the kernels that take the panel otherwise:
// gpurun_out/r3_narrow_shapes*.txt, DESIGN 3.9.
// (declared in kernels.h) A narrow call whose rows average less than three quarters of the bar does not classify at all: the classifier,
// an LDS-tiled launch that every workgroup leaves at once and the per-row ownership test of the direct kernel cost a
// 1 M-row matrix of 5 nonzeros per row 0.23 ms of a 0.09 ms product (N = 8).
static float window_min_rowlen(int64_t ldbt);
// what the classifier looks at for the direct kernels' sake (128+ staged columns): bit 0 rows that share column patterns
// (row-merging kernel), bit 1 column runs / row-length spread (row per wave or four rows per wave)
static int direct_probe(int64_t ldbt) { return ldbt >= 128 ? (options().direct_merge ? 3 : 2) : 0; }
bool classify_worthwhile(int64_t rows, int64_t nnz, int64_t ldbt)
{
    // (128+ staged columns: the classifier also feeds the matrix-core and row-merging choices; a forced matrix-core run
    //  needs its verdicts at any width)
    return ldbt >= 128 || (ldbt >= 64 && options().spmm_variant == SPMM_VARIANT_MFMA) || (rows > 0 && (double)nnz >= 0.75 * (double)window_min_rowlen(ldbt) * (double)rows);
}
static float window_min_rowlen(int64_t ldbt)
{
    const int t = options().tune[1];
    if (t > 1) return (float)t; /* SBLAS_TUNE=*,<nonzeros per row>: threshold sweeps */
    // 64+ columns, banded rows, 1 M rows, LDS-tiled kernel | four rows per wave (ms): N = 64: 5 per row .737 | .565, 16: .796 | .669,
    // 24: .849 | .804, 32: .924 | .944, 48: 1.06 | 1.22; N = 256: 16: 3.01 | 2.65, 24: 3.17 | 3.20, 32: 3.41 | 3.76
    if (ldbt >= 64) return 24.0f;
    // banded rows, 1 M rows, LDS-tiled kernel | lane groups (ms): N = 8: 27 per row .418 | .283, 40: .464 | .377, 60: .524 | .547,
    // 80: .466 | .554; N = 16: 10: .452 | .287, 27: .483 | .540; N = 32: 10: .672 | .553, 27: .711 | 1.04
    return ldbt <= 8 ? 56.0f : ldbt <= 16 ? 20.0f : 16.0f;
}
static float mfma_min_fill(int variant, int panel_rows, int64_t ldbt)
{
    if (panel_rows > 16 * MFMA_MAX_WAVES) return 2.0f; // the matrix-core kernel runs one wave per 16 rows of a panel
    if (variant == SPMM_VARIANT_MFMA) return 0.0f;
    if (variant == SPMM_VARIANT_NO_MFMA) return 2.0f;
    // Measured (tools/spmm_shapes.py): a chunk step of the matrix-core kernel costs ~2600 cycles of
    // instruction issue whatever the width, so it needs 128+ dense columns (8+ MFMAs per block) to pay: block-
    // structured rows at 60 % fill, N = 128: 0.57 ms against 0.65 ms for the LDS-tiled kernel; N = 64: 0.40 against
    // 0.31 ms; N = 256 at 35 % fill: 1.55 against 1.29 ms; grid-structured Queen-like rows (fill 0.32), N = 256: 2.33
    // against 2.16 ms for the direct kernel.  64-column calls therefore never take it unless SBLAS_MFMA_MIN_FILL says so.
    const float f = options().mfma_min_fill;
    if (f >= 0.0f) return f;
    return ldbt >= 128 ? 0.5f : 2.0f;
}

hipError_t launch_dense_to_rowmajor(hipStream_t s, int64_t cols, int64_t n, const double *B, int64_t ldb,
                                    double *Bt, int64_t ldbt)
{
    dim3 grid((unsigned)((cols + 1 + STAGE_K - 1) / STAGE_K), (unsigned)((ldbt + 63) / 64));
    int *hdr = reinterpret_cast<int *>(Bt + (size_t)(cols + 1) * (size_t)ldbt);
    const int epoch = g_epoch.fetch_add(1, std::memory_order_relaxed);
    const dim3 ngrid((unsigned)((cols + 1 + 255) / 256));
    if (ldbt == 8)
        hipLaunchKernelGGL(dense_to_rowmajor_narrow_kernel<8>, ngrid, dim3(256), 0, s, cols, n, B, ldb, Bt, hdr, epoch);
    else if (ldbt == 16)
        hipLaunchKernelGGL(dense_to_rowmajor_narrow_kernel<16>, ngrid, dim3(256), 0, s, cols, n, B, ldb, Bt, hdr, epoch);
    else if (ldbt == 32)
        hipLaunchKernelGGL(dense_to_rowmajor_narrow_kernel<32>, ngrid, dim3(256), 0, s, cols, n, B, ldb, Bt, hdr, epoch);
    else
        hipLaunchKernelGGL(dense_to_rowmajor_kernel, grid, dim3(256), 0, s, cols, n, B, ldb, Bt, ldbt, hdr, epoch);
    return hipGetLastError();
}

// Stage 1 of a row block: only the rows of B its nonzeros refer to.  With classify != 0 the panel classifier rides in
// the column-range launch and *epoch_out goes to launch_spmm_rowpanel; otherwise *epoch_out = 0.
hipError_t launch_stage_range(hipStream_t s, int64_t cols, int64_t n, const double *B, int64_t ldb, double *Bt,
                              int64_t ldbt, int rows, int64_t nnz, const int *rowptr, const int *colidx, int variant,
                              int classify, int *epoch_out)
{
    const Tail t = tail_of(Bt, cols, ldbt, rows);
    const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>((nnz + 4095) / 4096, TAIL_PARTS));
    // *epoch_out != 0 on entry: a later column chunk of the same call (same A, same ldbt, same workspace): the column
    // range and the panel verdicts in the tail still stand, only B's next columns need staging
    const bool again = *epoch_out != 0;
    const int epoch = again ? *epoch_out : g_epoch.fetch_add(1, std::memory_order_relaxed);
    if (!again && classify) {
        int info_rows = 0, g = 2;
        panel_plan(rows, ldbt, info_rows, g);
        const int np = (rows + info_rows - 1) / info_rows;
        hipLaunchKernelGGL(colrange_classify_kernel, dim3((unsigned)((np + 3) / 4 + nparts)), dim3(256), 0, s, nnz, t.parts,
                           nparts, rows, (int)cols, np, info_rows, rowptr, colidx, 1 << 24, window_min_density(info_rows), window_min_rowlen(ldbt),
                           mfma_min_fill(variant, info_rows, ldbt), direct_probe(ldbt), t.hdr,
                           t.info, t.cls, epoch);
    } else if (!again) {
        hipLaunchKernelGGL(colrange_kernel, dim3(nparts), dim3(256), 0, s, nnz, colidx, t.parts);
    }
    const int64_t tiles = ((cols + 1 + STAGE_K - 1) / STAGE_K) * ((ldbt + 63) / 64);
    // (a staging pass has an epoch of its own: "B holds a non-finite value" must not stick to the later column chunks of
    //  the call, which reuse the first chunk's classifier epoch)
    hipLaunchKernelGGL(stage_range_kernel, dim3((unsigned)std::min<int64_t>(tiles, 2048)), dim3(256), 0, s, cols, n, B,
                       ldb, Bt, ldbt, t.hdr, t.parts, nparts, g_epoch.fetch_add(1, std::memory_order_relaxed));
    *epoch_out = (classify || again) ? epoch : 0;
    return hipGetLastError();
}

// Stage 1 + classifier of the default path in one launch; the epoch goes to launch_spmm_rowpanel.
hipError_t launch_stage_classify(hipStream_t s, int64_t cols, int64_t n, const double *B, int64_t ldb, double *Bt,
                                 int64_t ldbt, int rows, const int *rowptr, const int *colidx, int variant,
                                 int *epoch_out)
{
    int info_rows = 0, g = 2;
    panel_plan(rows, ldbt, info_rows, g);
    const int np = (rows + info_rows - 1) / info_rows;
    const int stage_blocks = (int)((cols + 1 + STAGE_K - 1) / STAGE_K);
    const int epoch = g_epoch.fetch_add(1, std::memory_order_relaxed);
    const Tail t = tail_of(Bt, cols, ldbt, rows);
    if (ldbt < 64) {
        const dim3 ngrid((unsigned)((cols + 1 + 255) / 256 + (np + 3) / 4));
#define SBLAS_STAGE_NARROW(NC)                                                                                         \
    hipLaunchKernelGGL(stage_classify_narrow_kernel<NC>, ngrid, dim3(256), 0, s, cols, n, B, ldb, Bt, rows, np, info_rows, \
                       rowptr, colidx, 1 << 24, window_min_density(info_rows), window_min_rowlen(ldbt), t.hdr, t.info, t.cls, epoch)
        if (ldbt == 8) SBLAS_STAGE_NARROW(8);
        else if (ldbt == 16) SBLAS_STAGE_NARROW(16);
        else SBLAS_STAGE_NARROW(32);
#undef SBLAS_STAGE_NARROW
        *epoch_out = epoch;
        return hipGetLastError();
    }
    dim3 grid((unsigned)(stage_blocks + (np + 3) / 4), (unsigned)((ldbt + 63) / 64));
    hipLaunchKernelGGL(stage_classify_kernel, grid, dim3(256), 0, s, cols, n, B, ldb, Bt, ldbt, stage_blocks, rows, np,
                       info_rows, rowptr, colidx, 1 << 24, window_min_density(info_rows), window_min_rowlen(ldbt),
                       ldbt < 64 ? 2.0f : mfma_min_fill(variant, info_rows, ldbt),
                       direct_probe(ldbt), t.hdr, t.info, t.cls, epoch);
    *epoch_out = epoch;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Per-matrix plan (the slot cuSPARSE's bufferSize / preprocess step occupies, spmm.h:134-141): classify once, vote once,
// look at the verdicts once on the host.
// ---------------------------------------------------------------------------------------------
size_t plan_tail_bytes(int64_t rows) { return workspace_tail_bytes(rows); }

hipError_t plan_build(hipStream_t s, int rows, int cols, int64_t nnz, const int *rowptr, const int *colidx, int64_t ldbt,
                      int variant, bool use_range, PlanView *pv)
{
    const Tail t = tail_at(pv->tail, rows);
    int info_rows = 0, g = 2;
    panel_plan(rows, ldbt, info_rows, g);
    const int np = (rows + info_rows - 1) / info_rows;
    const int epoch = g_epoch.fetch_add(1, std::memory_order_relaxed);
    const float fill = ldbt < 64 ? 2.0f : mfma_min_fill(variant, info_rows, ldbt);
    const int probe = direct_probe(ldbt);
    hipError_t e = hipMemsetAsync(t.hdr, 0, TAIL_HDR * sizeof(int), s);
    if (e != hipSuccess) return e;
    const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>((nnz + 4095) / 4096, TAIL_PARTS));
    if (use_range)
        hipLaunchKernelGGL(colrange_classify_kernel, dim3((unsigned)((np + 3) / 4 + nparts)), dim3(256), 0, s, nnz, t.parts,
                           nparts, rows, cols, np, info_rows, rowptr, colidx, 1 << 24, window_min_density(info_rows), window_min_rowlen(ldbt), fill,
                           probe, t.hdr, t.info, t.cls, epoch);
    else
        hipLaunchKernelGGL(classify_panels_kernel, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, s, rows, cols, np, info_rows,
                           rowptr, colidx, 1 << 24, window_min_density(info_rows), window_min_rowlen(ldbt), fill, probe, t.hdr, t.info, t.cls, epoch);
    if (ldbt >= 128)
        hipLaunchKernelGGL(mfma_vote_kernel, dim3(1), dim3(1024), 0, s, np, t.hdr, t.info, t.cls, epoch,
                           variant == SPMM_VARIANT_MFMA ? 1 : 0);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // the one look the host takes
    std::vector<int> hdr(TAIL_HDR), cls(np);
    if ((e = hipMemcpyAsync(hdr.data(), t.hdr, TAIL_HDR * sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(cls.data(), t.cls, (size_t)np * sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    pv->epoch = epoch;
    pv->info_rows = info_rows;
    pv->groups = g;
    pv->n_window = pv->n_direct = pv->n_mfma_w = pv->n_mfma_d = 0;
    for (int p = 0; p < np; ++p) { // (a panel without nonzeros is PANEL_DIRECT: the direct kernel writes its beta * C)
        const int c = cls[p] & PANEL_CLASS_MASK;
        pv->n_window += c == PANEL_WINDOW;
        pv->n_direct += c == PANEL_DIRECT;
        pv->n_mfma_w += c == PANEL_MFMA_W;
        pv->n_mfma_d += c == PANEL_MFMA_D;
    }
    pv->merge = ldbt >= 128 && hdr[TAIL_MERGE_EPOCH] == epoch;
    pv->four_rows = ldbt >= 128 && !pv->merge && hdr[TAIL_ROWS_EPOCH] == epoch;
    pv->use_range = use_range;
    pv->nparts = nparts;
    return hipSuccess;
}

// stage 1 of a planned call: the flags "B holds a non-finite value" go to the PLAN's header (where the stage-2 kernels
// of the call look), the copy covers the plan's column range when the plan has one
hipError_t launch_stage_planned(hipStream_t s, int64_t cols, int64_t n, const double *B, int64_t ldb, double *Bt,
                                int64_t ldbt, const PlanView &pv)
{
    const int epoch = g_epoch.fetch_add(1, std::memory_order_relaxed);
    int *hdr = pv.tail;
    if (pv.use_range && ldbt >= 64) {
        const int64_t tiles = ((cols + 1 + STAGE_K - 1) / STAGE_K) * ((ldbt + 63) / 64);
        hipLaunchKernelGGL(stage_range_kernel, dim3((unsigned)std::min<int64_t>(tiles, 2048)), dim3(256), 0, s, cols, n, B, ldb,
                           Bt, ldbt, hdr, reinterpret_cast<const int2 *>(hdr + TAIL_HDR), pv.nparts, epoch);
        return hipGetLastError();
    }
    const dim3 ngrid((unsigned)((cols + 1 + 255) / 256));
    if (ldbt == 8)
        hipLaunchKernelGGL(dense_to_rowmajor_narrow_kernel<8>, ngrid, dim3(256), 0, s, cols, n, B, ldb, Bt, hdr, epoch);
    else if (ldbt == 16)
        hipLaunchKernelGGL(dense_to_rowmajor_narrow_kernel<16>, ngrid, dim3(256), 0, s, cols, n, B, ldb, Bt, hdr, epoch);
    else if (ldbt == 32)
        hipLaunchKernelGGL(dense_to_rowmajor_narrow_kernel<32>, ngrid, dim3(256), 0, s, cols, n, B, ldb, Bt, hdr, epoch);
    else
        hipLaunchKernelGGL(dense_to_rowmajor_kernel, dim3((unsigned)((cols + 1 + STAGE_K - 1) / STAGE_K), (unsigned)((ldbt + 63) / 64)),
                           dim3(256), 0, s, cols, n, B, ldb, Bt, ldbt, hdr, epoch);
    return hipGetLastError();
}

// the four-rows-per-wave direct kernel.  Workgroups of 4 waves (16 rows) for the shortest rows, 16 waves (64 rows) otherwise:
// 1 M banded rows of 5 / 10 / 20 / 32 per row, N = 64, 4 | 8 | 16 waves: 0.547 | 0.554 | 0.571, 0.632 | 0.613 | 0.617, 0.973 |
// 0.936 | 0.912, 1.22 | 1.17 | 1.14 ms; power-law rows averaging 3.2 (a 64-row workgroup waits for its longest row): 0.740 |
// 0.830 | 0.957 ms.  (SBLAS_TUNE=*,*,*,<4|8|16> pins the size: A/B runs)
static void launch_direct_rows(hipStream_t s, int rows, int cols, const int *rowptr, const int *colidx, const double *val,
                               const double *Bt, int64_t ldbt, int n, double alpha, double beta, double *C, int64_t ldc,
                               const int *hdr, const int *cls, int info_rows, int interleave, int epoch, int voted,
                               double avg_row)
{
    const int pin = options().tune[3];
    const int wv = pin == 4 || pin == 8 || pin == 16 ? pin : avg_row < 8.0 ? 4 : 16;
    const int rp = (rows + 4 * wv - 1) / (4 * wv);
    const dim3 grid((unsigned)rp, (unsigned)(ldbt / 64));
#define SBLAS_ROWS_GO(WV)                                                                                             \
    hipLaunchKernelGGL(spmm_direct_rows_kernel<WV>, grid, dim3(WV * 64), 0, s, rows, cols, rp, rowptr, colidx, val, Bt, \
                       ldbt, n, alpha, beta, C, ldc, hdr, cls, info_rows, interleave, epoch, voted)
    if (wv == 4) SBLAS_ROWS_GO(4);
    else if (wv == 8) SBLAS_ROWS_GO(8);
    else SBLAS_ROWS_GO(16);
#undef SBLAS_ROWS_GO
}

hipError_t launch_spmm_rowpanel(hipStream_t s, int rows, int cols, int64_t nnz, const int *rowptr, const int *colidx,
                                const double *val, const double *Bt, int64_t ldbt, int n, double alpha,
                                double beta, double *C, int64_t ldc, int variant, int pre_epoch, const PlanView *pv)
{
    const Options &opt = options();
    const double avg_row = rows > 0 ? (double)nnz / (double)rows : 0.0;
    const int dpp_long = opt.tune[2] > 0 ? opt.tune[2] : DPP_LONG; // (SBLAS_TUNE=*,*,<entries>: A/B runs of the long-row split)
    // a planned call (pv): the verdicts sit in the plan's buffer, nothing is classified or voted on, and only the kernels
    // that have panels are launched
    const bool need_window = !pv || pv->n_window + pv->n_mfma_w > 0;
    const bool need_mfma = !pv || pv->n_mfma_w + pv->n_mfma_d > 0;
    const bool need_direct = !pv || pv->n_direct + pv->n_mfma_d > 0;
    if (ldbt >= 64) {
        const Tail t = pv ? tail_at(pv->tail, rows) : tail_of(Bt, cols, ldbt, rows);
        const int *cls = nullptr;
        int info_rows = 1;
        // pre_epoch != 0: launch_stage_classify has classified the panels already
        // (64 staged columns and short rows throughout: everything goes to the four-rows-per-wave kernel unclassified)
        const bool classified = pv || (variant != SPMM_VARIANT_DIRECT_DPP && variant != SPMM_VARIANT_DIRECT_ROWS &&
                                       variant != SPMM_VARIANT_DIRECT_MERGE &&
                                       (pre_epoch != 0 || classify_worthwhile(rows, nnz, ldbt)));
        const bool preclassified = pv || (pre_epoch != 0 && classified);
        const int epoch = pv ? pv->epoch : preclassified ? pre_epoch : g_epoch.fetch_add(1, std::memory_order_relaxed);
        if (classified) {
            // 1. classify row panels; 2. LDS-tiled kernel and matrix-core kernel on the panels that qualify;
            // 3. direct kernel on the rest
            int gen6_g = 2;
            panel_plan(rows, ldbt, info_rows, gen6_g);
            if (pv) info_rows = pv->info_rows, gen6_g = pv->groups;
            const int np = (rows + info_rows - 1) / info_rows;
            if (!preclassified)
                hipLaunchKernelGGL(classify_panels_kernel, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, s, rows, cols, np,
                                   info_rows, rowptr, colidx, /* 32-bit buffer offsets inside a wave's rows */ 1 << 24,
                                   /* a (row, tile) visit costs what ~8 nonzeros cost in the direct kernel: ask for 8
                                      per row and 128-column tile on average */
                                   window_min_density(info_rows), window_min_rowlen(ldbt), mfma_min_fill(variant, info_rows, ldbt),
                                   direct_probe(ldbt), t.hdr, t.info, t.cls, epoch);
            const bool mfma_possible = mfma_min_fill(variant, info_rows, ldbt) <= 1.0f;
            // matrix-wide decisions before stage 2 (128+ staged columns only: 64-column calls have neither choice)
            if (ldbt >= 128 && !pv)
                hipLaunchKernelGGL(mfma_vote_kernel, dim3(1), dim3(1024), 0, s, np, t.hdr, t.info, t.cls, epoch,
                                   variant == SPMM_VARIANT_MFMA ? 1 : 0);
            // 128+ staged columns: two 64-column halves per workgroup, the selection work of a (rows, tile) visit shared
            // (SBLAS_TUNE=*,1 keeps one half per workgroup: A/B runs)
            const bool two_halves = ldbt >= 128 && opt.tune[1] != 1;
            dim3 wgrid((unsigned)np, (unsigned)(ldbt / (two_halves ? 128 : 64)));
            KernelEvents *kev = kernel_events_slot();
            if (kev) (void)hipEventRecord(kev->a, s);
#define SBLAS_LAUNCH_W6(GG, NH)                                                                                        \
    do {                                                                                                              \
        raise_dynamic_lds((const void *)spmm_window6_kernel<GG, NH>, W2_LDS_BYTES);                                   \
        hipLaunchKernelGGL((spmm_window6_kernel<GG, NH>), wgrid, dim3(1024), W2_LDS_BYTES, s, rows, cols, np, rowptr,   \
                           colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, t.hdr, t.info, t.cls, info_rows, (int)nnz); \
    } while (0)
            if (!need_window) {
            } else if (gen6_g == 3) {
                SBLAS_LAUNCH_W6(3, 1);
            } else {
                if (two_halves) SBLAS_LAUNCH_W6(2, 2); else SBLAS_LAUNCH_W6(2, 1);
            }
#undef SBLAS_LAUNCH_W6
            if (kev) {
                (void)hipEventRecord(kev->b, s);
                kev->recorded = true;
            }
            if (mfma_possible && need_mfma) {
                const hipError_t e = launch_spmm_mfma(s, rows, cols, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc,
                                                      t.info, t.hdr, t.cls, info_rows, np, epoch, panel_stats_device());
                if (e != hipSuccess) return e;
            }
            cls = t.cls;
        }
        const int wide_panels = (rows + WIDE_PANEL - 1) / WIDE_PANEL;
        // 128-column tiles: one workgroup per CU through unused dynamic LDS (Queen-like rows at N = 256: +13 %, banded
        // matrix at N = 128: +3 %); SBLAS_DIRECT_LDS overrides (experiments: rows in flight vs L2 reach)
        const size_t pad = opt.direct_lds >= 0 ? (size_t)opt.direct_lds : (ldbt == 64 ? 0 : 90000);
        const int interleave = opt.direct_map; // -1: by the span the classifier recorded
        if (!need_direct) {
        } else if (n > 32 && (variant == SPMM_VARIANT_DIRECT_ROWS ||
                              (variant != SPMM_VARIANT_DIRECT_DPP && avg_row < (ldbt == 64 ? 56.0 : 32.0)))) {
            // short rows: four rows per wave (64 staged columns, banded rows, 1 M rows, against the lane-group kernel: 32 per
            // row 1.08 | 1.14 ms, 48: 1.41 | 1.54, 64: 1.75 | 1.70, 100: 2.57 | 2.49)
            launch_direct_rows(s, rows, cols, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, t.hdr, cls, info_rows,
                               interleave, epoch, 0, avg_row);
        } else if (ldbt == 64 && n <= 32) {
            if (pad) raise_dynamic_lds((const void *)spmm_direct_dpp_kernel<4>, pad);
            hipLaunchKernelGGL(spmm_direct_dpp_kernel<4>, dim3((unsigned)wide_panels, 1u), dim3(WIDE_WAVES * 64), pad, s,
                               rows, cols, wide_panels, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, t.hdr, cls,
                               info_rows, interleave, epoch, dpp_long);
        } else if (ldbt == 64) {
            if (pad) raise_dynamic_lds((const void *)spmm_direct_dpp_kernel<2>, pad);
            hipLaunchKernelGGL(spmm_direct_dpp_kernel<2>, dim3((unsigned)wide_panels, 1u), dim3(WIDE_WAVES * 64), pad, s,
                               rows, cols, wide_panels, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, t.hdr, cls,
                               info_rows, interleave, epoch, dpp_long);
        } else {
            // 128-column tiles.  Classified calls launch both direct kernels: the classifier's vote (device side) says
            // whether the rows share column patterns, and the kernel whose call it is not leaves on one scalar load.
            const bool merge = pv ? pv->merge : variant == SPMM_VARIANT_DIRECT_MERGE || (cls != nullptr && opt.direct_merge);
            const bool plain = pv ? !pv->merge && !pv->four_rows : variant != SPMM_VARIANT_DIRECT_MERGE;
            // ... and whether four rows per wave on 64-column tiles suit them better than a row per wave on 128-column tiles
            const bool four = pv ? pv->four_rows : cls != nullptr && variant != SPMM_VARIANT_DIRECT_MERGE;
            if (four) {
                launch_direct_rows(s, rows, cols, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, t.hdr, cls, info_rows,
                                   interleave, epoch, pv ? 0 : 1, avg_row);
            }
            if (merge) {
                // rows that share their column pattern (multi-dof FEM): three rows per wave, shared Bt loads
                const int mp = (rows + MERGE_PANEL - 1) / MERGE_PANEL;
                const size_t lds = (size_t)128 * (MERGE_PANEL + 1) * sizeof(double);
                raise_dynamic_lds((const void *)spmm_direct_merge_kernel, lds);
                hipLaunchKernelGGL(spmm_direct_merge_kernel, dim3((unsigned)mp, (unsigned)(ldbt / 128)),
                                   dim3(MERGE_WAVES * 64), lds, s, rows, cols, mp, rowptr, colidx, val, Bt, ldbt, n, alpha,
                                   beta, C, ldc, t.hdr, cls, info_rows, interleave, epoch);
            }
            if (plain) {
                if (pad) raise_dynamic_lds((const void *)spmm_direct_dpp_kernel<1>, pad);
                hipLaunchKernelGGL(spmm_direct_dpp_kernel<1>, dim3((unsigned)wide_panels, (unsigned)(ldbt / 128)),
                                   dim3(WIDE_WAVES * 64), pad, s, rows, cols, wide_panels, rowptr, colidx, val, Bt, ldbt, n,
                                   alpha, beta, C, ldc, t.hdr, cls, info_rows, interleave, epoch, dpp_long);
            }
        }
    } else {
        // ---- narrow dense blocks (ldbt = 8 / 16 / 32): LDS-tiled lane-per-entry kernel on the panels that qualify,
        // a direct kernel on the rest
        const Tail t = pv ? tail_at(pv->tail, rows) : tail_of(Bt, cols, ldbt, rows);
        // the 16- / 32-column direct kernel addresses Bt with 32-bit byte offsets
        const bool wide_offsets = ((uint64_t)cols + 1) * (uint64_t)ldbt * 8ull > 0xffffffffull;
        const bool classified = pv || (!wide_offsets && variant != SPMM_VARIANT_DIRECT_DPP && variant != SPMM_VARIANT_DIRECT_ROWS &&
                                       variant != SPMM_VARIANT_LANES && (pre_epoch != 0 || classify_worthwhile(rows, nnz, ldbt)));
        const bool preclassified = pv || (pre_epoch != 0 && classified);
        const int epoch = pv ? pv->epoch : preclassified ? pre_epoch : g_epoch.fetch_add(1, std::memory_order_relaxed);
        const int *cls = nullptr;
        int info_rows = 1;
        if (classified) {
            int g = 2;
            lanes_plan(rows, (int)ldbt, info_rows, g);
            if (pv) info_rows = pv->info_rows, g = pv->groups;
            const int np = (rows + info_rows - 1) / info_rows;
            if (!preclassified)
                hipLaunchKernelGGL(classify_panels_kernel, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, s, rows, cols, np,
                                   info_rows, rowptr, colidx, 1 << 24, window_min_density(info_rows), window_min_rowlen(ldbt), 2.0f, 0, t.hdr, t.info,
                                   t.cls, epoch);
            KernelEvents *kev = kernel_events_slot();
            if (kev) (void)hipEventRecord(kev->a, s);
#define SBLAS_LAUNCH_LANES(NC, CP, GG) SBLAS_LAUNCH_LANES4(NC, CP, GG, 1)
#define SBLAS_LAUNCH_LANES4(NC, CP, GG, LPE)                                                                           \
    do {                                                                                                              \
        raise_dynamic_lds((const void *)spmm_lanes_kernel<NC, CP, GG, LPE>, WlGeom<NC, CP>::LDS_BYTES);               \
        hipLaunchKernelGGL((spmm_lanes_kernel<NC, CP, GG, LPE>), dim3((unsigned)np), dim3(1024), (WlGeom<NC, CP>::LDS_BYTES), s, \
                           rows, cols, np, rowptr, colidx, val, Bt, n, alpha, beta, C, ldc, t.hdr, t.info, t.cls,      \
                           info_rows, (int)nnz);                                                                      \
    } while (0)
            const int cp = opt.tune[0]; /* experiments: copies of a Bt row in the LDS tile */
            if (!need_window) {
            } else if (ldbt == 8) {
                if (cp == 4) { if (g == 3) SBLAS_LAUNCH_LANES(8, 4, 3); else if (g == 2) SBLAS_LAUNCH_LANES(8, 4, 2); else SBLAS_LAUNCH_LANES(8, 4, 1); }
                else if (cp == 2) { if (g == 3) SBLAS_LAUNCH_LANES(8, 2, 3); else if (g == 2) SBLAS_LAUNCH_LANES(8, 2, 2); else SBLAS_LAUNCH_LANES(8, 2, 1); }
                else { if (g == 3) SBLAS_LAUNCH_LANES(8, 1, 3); else if (g == 2) SBLAS_LAUNCH_LANES(8, 1, 2); else SBLAS_LAUNCH_LANES(8, 1, 1); }
            } else if (ldbt == 16) {
                if (cp == 2) { if (g == 2) SBLAS_LAUNCH_LANES(16, 2, 2); else SBLAS_LAUNCH_LANES(16, 2, 1); }
                else if (g == 3) SBLAS_LAUNCH_LANES4(16, 1, 3, 2);   // two lanes per entry: eight accumulators per lane and group
                else { if (g == 2) SBLAS_LAUNCH_LANES(16, 1, 2); else SBLAS_LAUNCH_LANES(16, 1, 1); }
            } else {
                // 32 columns: two lanes per entry (sixteen accumulators per lane: two groups of rows per wave fit), or a
                // lane per entry with one group (SBLAS_TUNE=*,*,*,1: A/B runs)
                if (g == 2) SBLAS_LAUNCH_LANES4(32, 1, 2, 2); else if (opt.tune[3] == 1) SBLAS_LAUNCH_LANES(32, 1, 1); else SBLAS_LAUNCH_LANES4(32, 1, 1, 2);
            }
#undef SBLAS_LAUNCH_LANES
#undef SBLAS_LAUNCH_LANES4
            if (kev) {
                (void)hipEventRecord(kev->b, s);
                kev->recorded = true;
            }
            cls = t.cls;
        }
        const unsigned panels = (unsigned)((rows + PANEL_ROWS - 1) / PANEL_ROWS);
        // short rows leave most of a row-per-wave sweep empty: lane groups below 24 / 16 nonzeros per row on average at 16 / 32
        // columns (banded rows of 5 / 10 per row, 1 M rows: N = 16 0.214 / 0.287 ms against 0.585 / 0.667 row per wave; N = 32
        // 0.403 / 0.553 against 0.691 / 0.758; from 27 per row on the row-per-wave kernel wins)
        const bool short_rows = ldbt >= 16 && avg_row < (ldbt == 16 ? 24.0 : 16.0) && variant != SPMM_VARIANT_DIRECT_DPP;
        if (!need_direct) {
        } else if (ldbt >= 16 && !wide_offsets && variant != SPMM_VARIANT_LANES && !short_rows) {
            // the row-per-wave kernel, four nonzeros per instruction: sixteen lanes x 16 bytes per nonzero (with 16 staged
            // columns the upper eight lanes of a DPP row read past the Bt row, into columns that are never stored)
            const int wide_panels = (rows + WIDE_PANEL - 1) / WIDE_PANEL;
            hipLaunchKernelGGL(spmm_direct_dpp_kernel<4>, dim3((unsigned)wide_panels, 1u), dim3(WIDE_WAVES * 64), 0, s, rows,
                               cols, wide_panels, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, t.hdr, cls, info_rows,
                               opt.direct_map, epoch, dpp_long);
        } else if (ldbt == 32) {
            hipLaunchKernelGGL(spmm_rowpanel_narrow_kernel<32>, dim3(panels), dim3(256), 0, s, rows, rowptr, colidx, val,
                               Bt, n, alpha, beta, C, ldc, t.hdr, cls, info_rows, epoch);
        } else if (ldbt == 16) {
            hipLaunchKernelGGL(spmm_rowpanel_narrow_kernel<16>, dim3(panels), dim3(256), 0, s, rows, rowptr, colidx, val,
                               Bt, n, alpha, beta, C, ldc, t.hdr, cls, info_rows, epoch);
        } else if (avg_row >= opt.rows8_min_avg && variant != SPMM_VARIANT_LANES) {
            // n <= 8, long rows: a wave per row, eight sums per lane (banded-random rows, band +-20000, 600 k rows,
            // N = 8: 64 / 128 / 200 / 300 per row: the lane groups win by 25 / 30 / 2 / 0 %; bench matrix, 399 per row,
            // band +-2000: the wave per row wins by 20 % -- tools/rows8_threshold.py)
            hipLaunchKernelGGL(spmm_rows8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, rows, cols, rowptr,
                               colidx, val, Bt, n, alpha, beta, C, ldc, t.hdr, cls, info_rows, epoch);
        } else {
            hipLaunchKernelGGL(spmm_rowpanel_narrow_kernel<8>, dim3(panels), dim3(256), 0, s, rows, rowptr, colidx, val,
                               Bt, n, alpha, beta, C, ldc, t.hdr, cls, info_rows, epoch);
        }
    }
    return hipGetLastError();
}

// device address of the panel census (the matrix-core kernel lives in another translation unit)
unsigned long long *panel_stats_device()
{
    static std::atomic<unsigned long long *> ptr[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    unsigned long long *q = ptr[dev].load(std::memory_order_relaxed);
    if (!q) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_panel_stats)) != hipSuccess) return nullptr;
        q = static_cast<unsigned long long *>(p);
        ptr[dev].store(q, std::memory_order_relaxed);
    }
    return q;
}

hipError_t panel_stats(unsigned long long out[4], bool reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_panel_stats), 4 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_panel_stats), z, sizeof z);
    }
    return e;
}


hipError_t launch_scale(hipStream_t s, int64_t rows, int64_t n, double beta, double *C, int64_t ldc)
{
    hipLaunchKernelGGL(scale_kernel, dim3(capped_grid(rows * n, 256)), dim3(256), 0, s, rows, n, beta, C, ldc);
    return hipGetLastError();
}

hipError_t launch_axpby(hipStream_t s, int64_t n, double alpha, const double *x, double beta, double *y)
{
    hipLaunchKernelGGL(axpby_kernel, dim3(capped_grid(n, 512)), dim3(256), 0, s, n, alpha, x, beta, y);
    return hipGetLastError();
}

hipError_t launch_merge_rowblocks(hipStream_t s, int64_t M, int64_t N, int g, const double *const *src,
                                  const int64_t *start, const int64_t *nrows, double alpha, double beta, double *C,
                                  int64_t ldc)
{
    RowBlocks b{};
    for (int q = 0; q < g; ++q) {
        b.src[q] = src[q];
        b.start[q] = start[q];
        b.nrows[q] = nrows[q];
    }
    hipLaunchKernelGGL(merge_rowblocks_kernel, dim3(capped_grid(M * N, 256)), dim3(256), 0, s, (long long)M,
                       (long long)N, g, b, alpha, beta, C, (long long)ldc);
    return hipGetLastError();
}

} // namespace sblas
