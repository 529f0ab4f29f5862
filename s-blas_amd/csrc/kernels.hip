// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the S-BLAS CSR hot path.
//
// SpMM  C = alpha*A*B + beta*C   (stage 1 + stage 2; the launcher at the end of this file picks the kernels)
//   dense_to_rowmajor_kernel     B (col-major) -> Bt (row-major, zero padded, one all-zero row)          [stage 1]
//   stage_classify_kernel        stage 1 + classify_panels_kernel in one launch (fused C-ABI entry, default variant)
//   classify_panels_kernel       per row panel: dense enough over its column span for the LDS-tiled kernel?
//   spmm_window6_kernel<G>       DEFAULT for qualifying panels: 128-row x 64-column B tiles through LDS (LDS-DMA
//                                loader waves), one DPP row per matrix row, streaming windows of A (generation 6)
//   spmm_direct_dpp_kernel<GROUPS> DEFAULT for all other panels: a row per wave, Bt rows straight from L2, DPP broadcast
//   spmm_direct_rows_kernel      direct panels of short-row matrices (< 32 per row): four rows per wave
//   spmm_rowpanel_narrow_kernel  n <= 8 (sub-wave lane groups; 16 / 32 columns behind SBLAS_SPMM_MIN_LDBT=0)
//   spmm_rows8_kernel            n <= 8 and rows of 256+ nonzeros on average: a wave per row, eight sums per lane
//   spmm_window{,2,3,4,5}_kernel, spmm_rowpanel_kernel   earlier generations, selectable (SBLAS_SPMM_VARIANT) and
//                                kept as regression cases of the parity suite
// SpMV  y = alpha*A*x + beta*y
//   spmv_csr_lds_kernel<RW,S>    DEFAULT above 96 nonzeros per row: x window of a 16-row block in LDS
//   spmv_csr_seg_kernel<R,S>     DEFAULT for 49..96 per row: R rows per wave, segmented reduction
//   spmv_csr_stream_kernel       DEFAULT for 6..48 per row: 256 rows per block streamed through LDS
//   spmv_csr_kernel<LPR>         DEFAULT up to 5 per row: LPR lanes per row, xor-shuffle fold
//   spmv_csr_{burst,window,flat}_kernel   experiments (SBLAS_SPMV_VARIANT)
// Epilogues and merges
//   axpby_kernel                 y = beta*y + alpha*x                                        (kernel.h:27-38)
//   merge_rowblocks_kernel       method-2 / SpMV merge: scatter packed row blocks, apply alpha / beta
//   sum_replicas_kernel          in-place sum over g buffers that live on ONE device (folded ranks)
//
// These replace the closed-source cuSPARSE calls of the reference (spmm.h:146-149, :248-251;
// spmv.h:104-106) and its one utility kernel (kernel.h:27-38).  Everything is written for
// 64-wide wavefronts; there is no 32-lane code path.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <string.h>
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
__device__ __forceinline__ void stage_tile(double (*tile)[STAGE_K + 1], int64_t k0, int64_t j0, int64_t cols, int64_t n,
                                           const double *__restrict__ B, int64_t ldb, double *__restrict__ Bt,
                                           int64_t ldbt)
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
                                                               double *__restrict__ Bt, int64_t ldbt)
{
    __shared__ double tile[64][STAGE_K + 1];
    stage_tile(tile, (int64_t)blockIdx.x * STAGE_K, (int64_t)blockIdx.y * 64, cols, n, B, ldb, Bt, ldbt);
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

__global__ __launch_bounds__(WIDE_WAVES * 64) void spmm_rowpanel_kernel(
    int rows, int npanels, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc)
{
    __shared__ double ctile[64][WIDE_PANEL + 1];
    const int lane = threadIdx.x & 63;
    const int wave = wave_uniform(threadIdx.x >> 6);
    const int row0 = xcd_contiguous_panel(blockIdx.x, npanels) * WIDE_PANEL;
    const int col0 = blockIdx.y * 64;
    // Row c of Bt starts at element c*ldbt: a 32-bit scalar product (cols*ldbt < 2^32 is checked by the
    // C ABI), so the load is "scalar row base + one constant per-lane offset" with no vector address math.
    const unsigned lane_off = (unsigned)(col0 + lane);
    const unsigned ld32 = (unsigned)ldbt;

    for (int rr = 0; rr < WIDE_ROWS_PER_WAVE; ++rr) {
        const int r = wave * WIDE_ROWS_PER_WAVE + rr;
        const int row = row0 + r;
        double acc = 0.0;
        if (row < rows) {
            const int p0 = wave_uniform(rowptr[row]);
            const int p1 = wave_uniform(rowptr[row + 1]);
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
        }
        ctile[lane][r] = acc;
    }
    __syncthreads();

    // write-back: consecutive threads walk consecutive rows of one column of C (128 B per column)
    const int nrows = min(WIDE_PANEL, rows - row0);
    const int ncols = min(64, n - col0);
    for (int idx = threadIdx.x; idx < 64 * WIDE_PANEL; idx += WIDE_WAVES * 64) {
        const int r = idx % WIDE_PANEL, j = idx / WIDE_PANEL;
        if (r < nrows && j < ncols) {
            double *dst = C + (int64_t)(col0 + j) * ldc + (row0 + r);
            const double s = alpha * ctile[j][r];
            *dst = (beta == 0.0) ? s : fma(beta, *dst, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 2, windowed form: row panel x dense B tile through LDS.
//
// The direct kernel above pulls one 512-byte Bt row per nonzero through the vector-memory path
// (64 B/clk/CU), i.e. >= 8 clk per nonzero per CU.  When the rows of a panel share a narrow column
// range (banded / well-ordered FEM matrices) the Bt rows can be staged ONCE per panel in LDS and each
// nonzero then costs one conflict-free ds_read_b64 per lane (256 B/clk/CU):
//
//   workgroup = 16 waves = a panel of R = 16*RPW rows x one 64-column tile of C;
//   the panel's column span [cmin, cmax] is walked in tiles of WIN_W Bt rows (64 KiB each, two LDS
//   buffers: tile t+1 is fetched into registers while tile t is consumed, and written to LDS after
//   it -- one barrier per tile);
//   every wave keeps, for each of its RPW rows, the current 64-nonzero chunk of (col, val) in
//   registers plus the next chunk prefetched; per tile it consumes the prefix of the chunk whose
//   columns fall inside the tile (rows with ascending columns: a ballot + popcount).
//
// Nothing is assumed about the input: a panel takes this path only if it is dense enough over its
// span to pay for the tile loads, every consumed nonzero is checked to lie inside the current tile,
// and a panel that breaks the ascending-column expectation (or ends with unconsumed nonzeros) is
// recomputed by the direct per-row loop before anything is written to C.
// ---------------------------------------------------------------------------------------------
// panel census of the windowed kernel: [0] windowed, [1] direct (too sparse over its span), [2] windowed but
// recomputed by the fallback.  One atomic per panel; read through sblas_hip_debug_spmm_panel_stats.
__device__ unsigned long long g_panel_stats[4];
// cycle stamps of the diagnostic mode (SBLAS_ABLATE bit 2): [0] consumer prologue, [1] consumer visits,
// [2] consumer barrier waits, [3] consumer epilogue, [4] loader put, [5] loader fetch issue, [6] loader barrier
// waits, [7] samples (consumer waves), [8] samples (loader waves), [9] whole kernel per wave
__device__ unsigned long long g_prof[16];
__device__ __forceinline__ unsigned long long stamp()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

constexpr int WIN_THREADS = 1024;
// two B tiles + one all-zero Bt row (target of masked DPP slots) + a few ints
constexpr size_t win_lds_bytes(int W) { return (2 * (size_t)W * 64 + 64) * sizeof(double) + 64 * sizeof(int); }

// Four nonzeros at DPP slots K0..K0+3 of every 16-lane row:
//   addr_k = row_newbcast:k(co) + lb      (co = byte offset of the nonzero's Bt row inside the tile)
//   acc   += row_newbcast:k(gv) * LDS[addr_k]
// v_add_u32_dpp / v_fmac_f64_dpp are full-rate VOP2 ops; v_readlane_b32 (the obvious broadcast) measured
// ~8 cycles per wave-instruction on gfx950 and made the kernel VALU-bound.  The leading s_nop covers the
// "VALU write -> DPP read" hazard on co / gv; the LDS reads are counted by hand inside the statement.
#define SBLAS_DPP4(K0, K1, K2, K3)                                                                                   \
    asm volatile("s_nop 1\n\t"                                                                                       \
                 "v_add_u32_dpp %[a0], %[co], %[lb] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "v_add_u32_dpp %[a1], %[co], %[lb] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "v_add_u32_dpp %[a2], %[co], %[lb] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "v_add_u32_dpp %[a3], %[co], %[lb] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "ds_read_b64 %[d0], %[a0]\n\t"                                                                      \
                 "ds_read_b64 %[d1], %[a1]\n\t"                                                                      \
                 "ds_read_b64 %[d2], %[a2]\n\t"                                                                      \
                 "ds_read_b64 %[d3], %[a3]\n\t"                                                                      \
                 "s_waitcnt lgkmcnt(3)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[acc], %[gv], %[d0] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"           \
                 "s_waitcnt lgkmcnt(2)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[acc], %[gv], %[d1] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"           \
                 "s_waitcnt lgkmcnt(1)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[acc], %[gv], %[d2] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"           \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[acc], %[gv], %[d3] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"           \
                 : [acc] "+v"(acc), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3),                  \
                   [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)                                    \
                 : [co] "v"(co), [lb] "v"(lb), [gv] "v"(gv)                                                          \
                 : "memory")

// Eight nonzeros at DPP slots K0..K0+7, fully pipelined: all eight LDS reads are in flight before the first
// FMA, and the FMAs alternate between two accumulators so that the fp64 dependency chain is half as long.
// The two address temporaries are recycled: a ds_read has consumed its address operand once it has issued.
#define SBLAS_DPP8(K0, K1, K2, K3, K4, K5, K6, K7)                                                                   \
    asm volatile("s_nop 1\n\t"                                                                                       \
                 "v_add_u32_dpp %[a0], %[co], %[lb] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "v_add_u32_dpp %[a1], %[co], %[lb] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "ds_read_b64 %[d0], %[a0]\n\t"                                                                      \
                 "ds_read_b64 %[d1], %[a1]\n\t"                                                                      \
                 "v_add_u32_dpp %[a0], %[co], %[lb] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "v_add_u32_dpp %[a1], %[co], %[lb] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "ds_read_b64 %[d2], %[a0]\n\t"                                                                      \
                 "ds_read_b64 %[d3], %[a1]\n\t"                                                                      \
                 "v_add_u32_dpp %[a0], %[co], %[lb] row_newbcast:" #K4 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "v_add_u32_dpp %[a1], %[co], %[lb] row_newbcast:" #K5 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "ds_read_b64 %[d4], %[a0]\n\t"                                                                      \
                 "ds_read_b64 %[d5], %[a1]\n\t"                                                                      \
                 "v_add_u32_dpp %[a0], %[co], %[lb] row_newbcast:" #K6 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "v_add_u32_dpp %[a1], %[co], %[lb] row_newbcast:" #K7 " row_mask:0xf bank_mask:0xf\n\t"             \
                 "ds_read_b64 %[d6], %[a0]\n\t"                                                                      \
                 "ds_read_b64 %[d7], %[a1]\n\t"                                                                      \
                 "s_waitcnt lgkmcnt(7)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c0], %[gv], %[d0] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "s_waitcnt lgkmcnt(6)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c1], %[gv], %[d1] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "s_waitcnt lgkmcnt(5)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c0], %[gv], %[d2] row_newbcast:" #K2 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "s_waitcnt lgkmcnt(4)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c1], %[gv], %[d3] row_newbcast:" #K3 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "s_waitcnt lgkmcnt(3)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c0], %[gv], %[d4] row_newbcast:" #K4 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "s_waitcnt lgkmcnt(2)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c1], %[gv], %[d5] row_newbcast:" #K5 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "s_waitcnt lgkmcnt(1)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c0], %[gv], %[d6] row_newbcast:" #K6 " row_mask:0xf bank_mask:0xf\n\t"            \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c1], %[gv], %[d7] row_newbcast:" #K7 " row_mask:0xf bank_mask:0xf\n\t"            \
                 : [c0] "+v"(acc), [c1] "+v"(acc_b), [a0] "=&v"(a0), [a1] "=&v"(a1), [d0] "=&v"(d0), [d1] "=&v"(d1), \
                   [d2] "=&v"(d2), [d3] "=&v"(d3), [d4] "=&v"(d4), [d5] "=&v"(d5), [d6] "=&v"(d6), [d7] "=&v"(d7)    \
                 : [co] "v"(co), [lb] "v"(lb), [gv] "v"(gv)                                                          \
                 : "memory")

// Consume nonzeros [g0, g0+ng) (ng <= 16) of a 64-wide register chunk against the LDS tile whose first Bt row
// is tile_lo.  The segment is first copied into slots 0..ng-1 of every 16-lane DPP row (ds_bpermute, no LDS
// traffic); slots >= ng get value 0 and the address of the all-zero row, so they add exactly 0.
// acc / acc_b: even / odd slots (the caller adds them once per row).
__device__ __forceinline__ void consume_dpp16(double &acc, double &acc_b, int cj, double vj, int g0, int ng,
                                              int tile_lo, unsigned lb, unsigned zero_rel, int lane)
{
    const int sub = lane & 15;
    const int src = (g0 + sub) << 2;
    const int gc = __builtin_amdgcn_ds_bpermute(src, cj);
    const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(vj));
    const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(vj));
    const bool on = sub < ng;
    const unsigned co = on ? ((unsigned)(gc - tile_lo) << 9) : zero_rel;
    const double gv = on ? __hiloint2double(hi, lo) : 0.0;
    unsigned a0, a1, a2, a3;
    double d0, d1, d2, d3, d4, d5, d6, d7;
    if (ng > 4) {
        SBLAS_DPP8(0, 1, 2, 3, 4, 5, 6, 7);
        if (ng > 8) {
            if (ng > 12) {
                SBLAS_DPP8(8, 9, 10, 11, 12, 13, 14, 15);
            } else {
                SBLAS_DPP4(8, 9, 10, 11);
            }
        }
    } else {
        SBLAS_DPP4(0, 1, 2, 3);
    }
}

__device__ __forceinline__ void load_chunk(const int *__restrict__ colidx, const double *__restrict__ val, int p,
                                           int pend, int lane, int &c, double &v)
{
    const int idx = p + lane;
    c = 0x7fffffff; // lanes past the row end never compare below a tile bound
    v = 0.0;
    if (idx < pend) {
        c = colidx[idx];
        v = val[idx];
    }
}

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

template <int RPW, int WIN_W, int MIN_WAVES>
__global__ __launch_bounds__(WIN_THREADS, MIN_WAVES) void spmm_window_kernel(
    int rows, int cols, int npanels, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, float min_density)
{
    constexpr int R = 16 * RPW;
    constexpr int WIN_TILE = WIN_W * 64;        // doubles per tile
    constexpr int STAGE = WIN_TILE / 2 / WIN_THREADS; // 16-byte pieces per thread per tile
    static_assert(STAGE >= 1, "tile too small for the block");
    static_assert(64 * (R + 1) <= 2 * WIN_TILE, "C tile must fit in the (dead) B tile buffers");
    static_assert(WIN_W * 512 < (1 << 24), "tile byte offsets must fit the DPP address add");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *zero_row = smem + 2 * WIN_TILE;                              // 64 zeros
    int *sm_i = reinterpret_cast<int *>(smem + 2 * WIN_TILE + 64);        // [0]=cmin [1]=cmax [2]=bad

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    const int row0 = xcd_contiguous_panel(blockIdx.x, npanels) * R;
    const int col0 = blockIdx.y * 64;
    const unsigned lane_off = (unsigned)(col0 + lane);
    const unsigned ld32 = (unsigned)ldbt;

    if (tid == 0) {
        sm_i[0] = 0x7fffffff;
        sm_i[1] = -1;
        sm_i[2] = 0;
    }
    if (tid < 64) zero_row[tid] = 0.0;
    __syncthreads();
    if (tid < R) {
        const int row = row0 + tid;
        if (row < rows) {
            const int a = rowptr[row], b = rowptr[row + 1];
            if (b > a) {
                atomicMin(&sm_i[0], colidx[a]);     // first / last column: the span when columns ascend;
                atomicMax(&sm_i[1], colidx[b - 1]); // anything else is caught by the in-tile check below
            }
        }
    }
    // this wave's rows
    int p0[RPW], p1[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + wave * RPW + r;
        p0[r] = p1[r] = 0;
        if (row < rows) {
            p0[r] = wave_uniform(rowptr[row]);
            p1[r] = wave_uniform(rowptr[row + 1]);
        }
    }
    __syncthreads();
    const int cmin = sm_i[0], cmax = sm_i[1];
    const int last_row = min(row0 + R, rows);
    const int panel_nnz = wave_uniform(rowptr[last_row]) - wave_uniform(rowptr[row0]);
    bool windowed = cmax >= cmin && cmin >= 0 && cmax < cols &&
                    (float)panel_nnz >= min_density * (float)(cmax - cmin + 1);

    double acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r] = 0.0;

    if (windowed) {
        const int t_lo = cmin / WIN_W, t_hi = cmax / WIN_W;
        // per-row chunk state (registers; all indices static after unrolling)
        int base[RPW], pos[RPW];
        int cj[RPW], cn[RPW];
        double vj[RPW], vn[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            base[r] = p0[r];
            pos[r] = 0;
            load_chunk(colidx, val, p0[r], p1[r], lane, cj[r], vj[r]);
            load_chunk(colidx, val, p0[r] + WAVE, p1[r], lane, cn[r], vn[r]);
        }
        int bad = 0;

        // tile staging: 4 x 16 B per thread; chunk q -> Bt row q/32, 16-byte piece q%32
        double2 stage[STAGE];
        auto stage_load = [&](int t) {
#pragma unroll
            for (int i = 0; i < STAGE; ++i) {
                const int q = tid + WIN_THREADS * i;
                const int brow = t * WIN_W + (q >> 5);
                double2 x = make_double2(0.0, 0.0);
                if (brow < cols)
                    x = *reinterpret_cast<const double2 *>(Bt + (size_t)((unsigned)brow * ld32) + col0 + ((q & 31) << 1));
                stage[i] = x;
            }
        };
        auto stage_store = [&](int buf) {
#pragma unroll
            for (int i = 0; i < STAGE; ++i) {
                const int q = tid + WIN_THREADS * i;
                *reinterpret_cast<double2 *>(smem + buf * WIN_TILE + (q << 1)) = stage[i];
            }
        };

        stage_load(t_lo);
        stage_store(0);
        for (int t = t_lo; t <= t_hi; ++t) {
            const int cur = (t - t_lo) & 1;
            if (t < t_hi) stage_load(t + 1);
            __syncthreads(); // tile t is in LDS; nobody still reads the other buffer
            const int tile_lo = t * WIN_W, tile_hi = tile_lo + WIN_W;
            // LDS byte address of this lane's column in row 0 of the current tile, and the zero row relative to it
            const unsigned tile_base = (unsigned)(uintptr_t)(smem + cur * WIN_TILE);
            const unsigned lb = tile_base + (unsigned)lane * 8u;
            const unsigned zero_rel = (unsigned)(uintptr_t)zero_row - tile_base;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                for (;;) {
                    const int cnt = min(WAVE, p1[r] - base[r]);
                    if (pos[r] >= cnt) {
                        if (base[r] + WAVE >= p1[r]) break; // row finished
                        base[r] += WAVE;                     // next chunk (already in registers)
                        pos[r] = 0;
                        cj[r] = cn[r];
                        vj[r] = vn[r];
                        load_chunk(colidx, val, base[r] + WAVE, p1[r], lane, cn[r], vn[r]);
                        continue;
                    }
                    const bool live = lane >= pos[r] && lane < cnt;
                    const unsigned long long m = __ballot(live && cj[r] < tile_hi);
                    const int take = __popcll(m);
                    if (take == 0) break; // next nonzero belongs to a later tile
                    // the taken lanes must be exactly pos..pos+take-1 and lie inside this tile
                    const unsigned long long want = ((take == 64) ? ~0ull : ((1ull << take) - 1ull)) << pos[r];
                    const unsigned long long below = __ballot(live && cj[r] < tile_lo);
                    if (m != want || below != 0ull) {
                        bad = 1;
                        pos[r] = cnt;
                        base[r] = p1[r]; // park the row; the panel will be recomputed
                        break;
                    }
                    const int k_end = pos[r] + take;
                    double a_acc = acc[r], b_acc = 0.0;
                    for (int g0 = pos[r]; g0 < k_end; g0 += 16)
                        consume_dpp16(a_acc, b_acc, cj[r], vj[r], g0, min(16, k_end - g0), tile_lo, lb, zero_rel, lane);
                    acc[r] = a_acc + b_acc;
                    pos[r] = k_end;
                    if (k_end < cnt) break; // chunk not exhausted: the rest is for later tiles
                }
            }
            if (t < t_hi) stage_store(cur ^ 1);
        }
        // every nonzero of every row must have been consumed
#pragma unroll
        for (int r = 0; r < RPW; ++r)
            if (base[r] + pos[r] < p1[r]) bad = 1;
        if (bad && lane == 0) atomicOr(&sm_i[2], 1);
        __syncthreads(); // also: all tile reads are done, the buffers may be reused for the C tile
        if (sm_i[2] != 0) {
            windowed = false;
#pragma unroll
            for (int r = 0; r < RPW; ++r) acc[r] = 0.0;
        }
    }
    if (tid == 0 && blockIdx.y == 0) atomicAdd(&g_panel_stats[windowed ? 0 : (sm_i[2] != 0 ? 2 : 1)], 1ull);
    if (!windowed) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[r] = row_direct(colidx, val, Bt, ld32, lane_off, lane, p0[r], p1[r]);
    }

    // park the panel as [column][row] (aliases the tile buffers, now dead) and write it back along rows
    double *ctile = smem;
#pragma unroll
    for (int r = 0; r < RPW; ++r) ctile[lane * (R + 1) + wave * RPW + r] = acc[r];
    __syncthreads();
    const int nrows = min(R, rows - row0);
    const int ncols = min(64, n - col0);
    for (int idx = tid; idx < 64 * R; idx += WIN_THREADS) {
        const int r = idx % R, j = idx / R;
        if (r < nrows && j < ncols) {
            double *dst = C + (int64_t)(col0 + j) * ldc + (row0 + r);
            const double s = alpha * ctile[j * (R + 1) + r];
            *dst = (beta == 0.0) ? s : fma(beta, *dst, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 2, windowed form, second generation: loader / consumer wave specialisation.
//
// The first windowed kernel lets every wave fetch its share of the next B tile AND stream its rows'
// (col,val) chunks; because vmcnt retires in order and hipcc waits vmcnt(0) at each use, every (row,tile)
// visit stalls on the tile prefetch issued a moment earlier (SQ_WAIT_ANY 58 %).  Here the roles are split:
//   waves 12..15 (loaders)   own ALL Bt traffic: tile t+1 -> registers -> the idle LDS buffer while the
//                            consumers work on tile t; they do nothing else;
//   waves 0..11 (consumers)  load the nonzeros of their RPW rows ONCE per panel into registers
//                            (CH chunks of 64 per row; rows longer than 64*CH send the panel to the direct
//                            kernel), check that each row's columns ascend, and then run the tile loop with
//                            LDS reads and DPP math only -- no global load, no vmcnt wait.
// One workgroup barrier per tile.  Which panels qualify is decided by classify_panels_kernel (below); the
// direct DPP kernel skips those panels and handles the rest, so any matrix is covered by the pair.
// ---------------------------------------------------------------------------------------------
constexpr int W2_ROWS = 128;              // Bt rows per LDS tile
constexpr int W2_TILE = W2_ROWS * 64;     // doubles
constexpr int W2_NCONS = 12; // consumer waves; the remaining 4 of the 16 waves are loaders
constexpr int W2_RPW = 4;
constexpr int W2_PANEL = W2_NCONS * W2_RPW; // 48 rows
constexpr size_t W2_LDS_BYTES = (2 * (size_t)W2_TILE + 64) * sizeof(double) + 64 * sizeof(int);

// info[p] = (cmin, cmax) of panel p when it should take the windowed path, (1, 0) otherwise.
// One wave per panel, one lane per row.
__device__ __forceinline__ void classify_panel(int p, int rows, int cols, int npanels, int panel_rows,
                                               const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                               int max_row_len, float min_density, int2 *__restrict__ info,
                                               int exclude_tail, int epoch)
{
    const int lane = threadIdx.x & 63;
    if (p >= npanels) return;
    int first = 0x7fffffff, last = -1, len = 0, mlen = 0;
    for (int rr = lane; rr < panel_rows; rr += WAVE) { // panels of up to 128 rows: two rows per lane
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
    if (lane == 0) {
        // exclude_tail: the panel that holds the very last nonzero goes to the direct kernel (fifth generation: its
        // col_idx fetch is eight bytes wide and must not run past the end of the array)
        const bool tail = exclude_tail && rowptr[min((p + 1) * panel_rows, rows)] == rowptr[rows];
        const bool ok = last >= first && first >= 0 && last < cols && mlen <= max_row_len && !tail &&
                        (float)nnz >= min_density * (float)(last - first + 1);
        // windowed: (first, last).  Direct: x > y, and for a non-empty panel the span stays recoverable
        // (first = -1 - x, last = -2 - y): the direct kernel samples three panels to choose its panel -> XCD map
        info[p] = ok ? make_int2(first, last) : (last >= first ? make_int2(-1 - first, -2 - last) : make_int2(1, 0));
        // the middle panel's column span, copied to the slot after the verdicts (one writer): the direct kernel takes
        // it as the band width of the matrix when it chooses its panel -> XCD map
        if (p == npanels / 2) info[npanels] = make_int2(last >= first ? last - first + 1 : 0, 0);
        // "this call left work for the direct kernel": the launch's epoch in the second spare slot (every writer
        // stores the same value).  The direct kernel leaves at once when the slot holds anything else -- an
        // optimisation only: a stale or accidental match merely sends it through its per-panel checks.
        if (!ok) info[npanels + 1].x = epoch;
    }
}
__global__ __launch_bounds__(256) void classify_panels_kernel(int rows, int cols, int npanels, int panel_rows,
                                                             const int *__restrict__ rowptr,
                                                             const int *__restrict__ colidx, int max_row_len,
                                                             float min_density, int2 *__restrict__ info,
                                                             int exclude_tail, int epoch)
{
    classify_panel(blockIdx.x * 4 + (threadIdx.x >> 6), rows, cols, npanels, panel_rows, rowptr, colidx, max_row_len,
                   min_density, info, exclude_tail, epoch);
}
// Stage 1 and the panel classifier in one launch (the fused C-ABI entry: both depend only on the call's inputs, and
// the classifier's few dependent loads hide behind the staging traffic): the first ceil(npanels / 4) workgroups
// (grid.y == 0 only) classify four panels each, the stage_blocks x grid.y behind them transpose B.
__global__ __launch_bounds__(256) void stage_classify_kernel(int64_t cols, int64_t n, const double *__restrict__ B,
                                                            int64_t ldb, double *__restrict__ Bt, int64_t ldbt,
                                                            int stage_blocks, int rows, int npanels, int panel_rows,
                                                            const int *__restrict__ rowptr,
                                                            const int *__restrict__ colidx, int max_row_len,
                                                            float min_density, int2 *__restrict__ info, int epoch)
{
    __shared__ double tile[64][STAGE_K + 1];
    // the classifier's workgroups come first in the grid: their chain of dependent loads starts at once and ends
    // under the staging traffic (placed last they stuck out by ~3 us)
    const int cblocks = (npanels + 3) / 4;
    if ((int)blockIdx.x >= cblocks) {
        stage_tile(tile, (int64_t)((int)blockIdx.x - cblocks) * STAGE_K, (int64_t)blockIdx.y * 64, cols, n, B, ldb, Bt,
                   ldbt);
    } else if (blockIdx.y == 0) {
        classify_panel((int)blockIdx.x * 4 + (threadIdx.x >> 6), rows, (int)cols, npanels, panel_rows, rowptr, colidx,
                       max_row_len, min_density, info, 0, epoch);
    }
}
// column span of a classified panel (0 for an empty one)
__device__ __forceinline__ int panel_span(int2 v)
{
    if (v.x >= 0) return v.y >= v.x ? v.y - v.x + 1 : 0;
    return (-2 - v.y) - (-1 - v.x) + 1;
}

template <int CH>
__global__ __launch_bounds__(1024) void spmm_window2_kernel(
    int rows, int cols, int npanels, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, const int2 *__restrict__ info, int ablate)
{
    constexpr int RPW = W2_RPW, R = W2_PANEL;
    static_assert(64 * (R + 1) <= 2 * W2_TILE, "C tile must fit in the (dead) B tile buffers");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *zero_row = smem + 2 * W2_TILE;
    int *sm_i = reinterpret_cast<int *>(smem + 2 * W2_TILE + 64); // [0] = bad

    const int panel = xcd_contiguous_panel(blockIdx.x, npanels);
    const int2 span = info[panel];
    if (span.x > span.y) return; // not a windowed panel: the direct kernel owns it (whole workgroup leaves)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    const int row0 = panel * R;
    const int col0 = blockIdx.y * 64;
    const unsigned ld32 = (unsigned)ldbt;
    const int t_lo = span.x / W2_ROWS, t_hi = span.y / W2_ROWS;
    const bool loader = wave >= W2_NCONS;

    if (tid < 64) zero_row[tid] = 0.0;
    if (tid == 0) sm_i[0] = 0;

    double acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r] = 0.0;

    if (loader) {
        // ---------------- loader waves: 256 threads move one 64 KiB tile = 16 x 16 B each ----------------
        const int ltid = tid - W2_NCONS * 64;
        double2 st[16];
        auto fetch = [&](int t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int q = ltid + 256 * i;
                const int brow = t * W2_ROWS + (q >> 5);
                double2 x = make_double2(0.0, 0.0);
                if (brow < cols)
                    x = *reinterpret_cast<const double2 *>(Bt + (size_t)((unsigned)brow * ld32) + col0 + ((q & 31) << 1));
                st[i] = x;
            }
        };
        auto put = [&](int buf) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int q = ltid + 256 * i;
                *reinterpret_cast<double2 *>(smem + buf * W2_TILE + (q << 1)) = st[i];
            }
        };
        const bool prof = (ablate & 4) != 0;
        unsigned long long T0 = 0, tput = 0, tfetch = 0, tbar = 0, ta = 0, tb = 0;
        if (prof) T0 = stamp();
        fetch(t_lo);
        put(0);
        if (t_lo < t_hi) fetch(t_lo + 1); // stays in flight across the barrier
        __syncthreads(); // P: tile t_lo is in LDS, the consumers' rows are in registers and validated
        if (sm_i[0] == 0) {
            for (int t = t_lo; t <= t_hi; ++t) {
                if (prof) ta = stamp();
                if (t < t_hi) {
                    // registers hold tile t+1 (fetched one iteration ago): park it in the buffer the consumers
                    // left at the previous barrier, then start fetching tile t+2 so that its L2 latency spans the
                    // consumers' whole next iteration
                    put(((t - t_lo) & 1) ^ 1);
                    if (prof) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tb = stamp(); tput += tb - ta; }
                    if (t + 1 < t_hi) fetch(t + 2);
                    if (prof) { ta = stamp(); tfetch += ta - tb; }
                }
                if (!(ablate & 1)) __syncthreads(); // E_t
                if (prof) { tb = stamp(); tbar += tb - ta; }
            }
        }
        if (prof && lane == 0) {
            atomicAdd(&g_prof[4], tput);
            atomicAdd(&g_prof[5], tfetch);
            atomicAdd(&g_prof[6], tbar);
            atomicAdd(&g_prof[8], 1ull);
            atomicAdd(&g_prof[9], stamp() - T0);
        }
        __syncthreads(); // V
    } else {
        // ---------------- consumer waves ----------------
        const bool prof = (ablate & 4) != 0;
        unsigned long long T0 = 0, tvis = 0, tbar = 0, ta = 0, tb = 0;
        if (prof) T0 = stamp();
        int p0[RPW], len[RPW];
        int cc[RPW];            // current chunk (columns / values), the one being consumed
        double cv[RPW];
        int sc[RPW][CH > 1 ? CH - 1 : 1]; // chunks 1..CH-1, waiting their turn
        double sv[RPW][CH > 1 ? CH - 1 : 1];
        int bad = 0;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = row0 + wave * RPW + r;
            p0[r] = 0;
            len[r] = 0;
            if (row < rows) {
                p0[r] = wave_uniform(rowptr[row]);
                len[r] = wave_uniform(rowptr[row + 1]) - p0[r];
            }
        }
        // every nonzero of my rows, issued back to back (one long burst per panel, then no global loads)
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            load_chunk(colidx, val, p0[r], p0[r] + len[r], lane, cc[r], cv[r]);
#pragma unroll
            for (int q = 1; q < CH; ++q)
                load_chunk(colidx, val, p0[r] + 64 * q, p0[r] + len[r], lane, sc[r][q - 1], sv[r][q - 1]);
        }
        // Nothing above is waited for here: chunk 0 is needed first (the compiler waits for exactly those two
        // loads), the stored chunks land while the first tiles are being processed.  Each chunk is checked for
        // ascending columns (equal neighbours allowed; lanes past the end hold INT_MAX) when it becomes current.
        auto chunk_ok = [&](int c, int cnt, int prev_last) -> bool {
            const int prev = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) << 2, c);
            return __ballot(lane > 0 && lane < cnt && c < prev) == 0ull && __builtin_amdgcn_readlane(c, 0) >= prev_last;
        };
        int last_col[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int cnt = min(64, len[r]);
            last_col[r] = -1;
            if (cnt > 0) {
                if (!chunk_ok(cc[r], cnt, -1)) bad = 1;
                last_col[r] = __builtin_amdgcn_readlane(cc[r], cnt - 1);
            }
        }
        if (bad && lane == 0) atomicOr(&sm_i[0], 1);
        __syncthreads(); // P
        if (prof && lane == 0) atomicAdd(&g_prof[0], stamp() - T0);
        if (sm_i[0] == 0) {
            int qcur[RPW], pos[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) qcur[r] = 0, pos[r] = 0;
            for (int t = t_lo; t <= t_hi; ++t) {
                if (prof) ta = stamp();
                const int cur = (t - t_lo) & 1;
                const int tile_lo = t * W2_ROWS, tile_hi = tile_lo + W2_ROWS;
                const unsigned tile_base = (unsigned)(uintptr_t)(smem + cur * W2_TILE);
                const unsigned lb = tile_base + (unsigned)lane * 8u;
                const unsigned zero_rel = (unsigned)(uintptr_t)zero_row - tile_base;
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    for (;;) {
                        const int cnt = min(64, len[r] - 64 * qcur[r]);
                        if (cnt <= 0) break; // row finished
                        const int take = __popcll(__ballot(lane >= pos[r] && lane < cnt && cc[r] < tile_hi));
                        if (take == 0) break; // next nonzero belongs to a later tile
                        const int k_end = pos[r] + take;
                        double a_acc = acc[r], b_acc = 0.0;
                        if (!(ablate & 2)) {
                            for (int g0 = pos[r]; g0 < k_end; g0 += 16)
                                consume_dpp16(a_acc, b_acc, cc[r], cv[r], g0, min(16, k_end - g0), tile_lo, lb,
                                              zero_rel, lane);
                        }
                        acc[r] = a_acc + b_acc;
                        pos[r] = k_end;
                        if (k_end < cnt) break;
                        // chunk exhausted: bring the next stored chunk forward (rare: once per 64 nonzeros)
                        qcur[r] += 1;
                        pos[r] = 0;
#pragma unroll
                        for (int q = 1; q < CH; ++q)
                            if (qcur[r] == q) {
                                cc[r] = sc[r][q - 1];
                                cv[r] = sv[r][q - 1];
                            }
                        const int ncnt = min(64, len[r] - 64 * qcur[r]);
                        if (ncnt > 0) {
                            if (!chunk_ok(cc[r], ncnt, last_col[r])) bad = 1;
                            last_col[r] = __builtin_amdgcn_readlane(cc[r], ncnt - 1);
                        }
                    }
                }
                if (prof) { tb = stamp(); tvis += tb - ta; }
                if (!(ablate & 1)) __syncthreads(); // E_t
                if (prof) { ta = stamp(); tbar += ta - tb; }
            }
            if (prof && lane == 0) {
                atomicAdd(&g_prof[1], tvis);
                atomicAdd(&g_prof[2], tbar);
                atomicAdd(&g_prof[7], 1ull);
                atomicAdd(&g_prof[9], stamp() - T0);
            }
            // every nonzero must have been consumed; a late chunk may have failed its order check
#pragma unroll
            for (int r = 0; r < RPW; ++r)
                if (64 * qcur[r] + pos[r] < len[r]) bad = 1;
            if (bad && lane == 0) atomicOr(&sm_i[0], 1);
        }
        __syncthreads(); // V: verdict of the whole panel
        if (sm_i[0] != 0) {
            // a row of this panel is not in ascending column order: recompute the panel straight from L2
            const unsigned lane_off = (unsigned)(col0 + lane);
#pragma unroll
            for (int r = 0; r < RPW; ++r)
                acc[r] = row_direct(colidx, val, Bt, ld32, lane_off, lane, p0[r], p0[r] + len[r]);
        }
    }
    if (tid == 0 && blockIdx.y == 0) atomicAdd(&g_panel_stats[sm_i[0] == 0 ? 0 : 2], 1ull);

    // park the panel as [column][row] in the (now dead) tile buffers and write it back along rows
    double *ctile = smem;
    if (!loader) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) ctile[lane * (R + 1) + wave * RPW + r] = acc[r];
    }
    __syncthreads(); // F
    const int nrows = min(R, rows - row0);
    const int ncols = min(64, n - col0);
    for (int idx = tid; idx < 64 * R; idx += 1024) {
        const int r = idx % R, j = idx / R;
        if (r < nrows && j < ncols) {
            double *dst = C + (int64_t)(col0 + j) * ldc + (row0 + r);
            const double sres = alpha * ctile[j * (R + 1) + r];
            *dst = (beta == 0.0) ? sres : fma(beta, *dst, sres);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 2, windowed form, third generation ("quad" consumer).
//
// Cycle stamps on the second generation showed the tile loop bound by VALU issue: ~5 vector instructions per
// nonzero (alignment, masks, one address add and one FMA per nonzero) at ~4 cycles each.  Here a consumer wave
// handles FOUR nonzeros per step, one per 16-lane DPP row, each lane owning four columns of the 64-column tile:
//     addr   = row_newbcast:k(co) + lane_base                 1 v_add_u32_dpp   (four different Bt rows at once)
//     d0, d1 = LDS[addr], LDS[addr + 256]                     2 ds_read_b128    (conflict-free: 16 lanes x 16 B)
//     acc0..3 += row_newbcast:k(val) * d                      4 v_fmac_f64_dpp
// i.e. 1.25 vector instructions per nonzero.  The register window of a row is stored in "quad order" (lane
// 16q+k holds entry 4k+q) and is kept LEFT-ALIGNED: after a visit has consumed `take` entries the window is
// shifted by `take` with ds_bpermute -- issued after the math, so its latency is hidden behind the other rows --
// and every visit starts at step 0 with the simple mask "entry < take".  The four DPP rows of a lane column
// hold partial sums over different nonzeros and are folded once per row at the end of the panel.
// Loader waves, tile protocol, classifier and fallback are those of the second generation.
// ---------------------------------------------------------------------------------------------
constexpr int W3_RPW = 3;
constexpr int W3_PANEL = W2_NCONS * W3_RPW; // 36 rows

// two steps (eight nonzeros), reads of both in flight before the first FMA; v110..v127 are scratch owned by
// the statement (named literally because the two halves of a 128-bit destination must be addressed separately)
#define SBLAS_QSTEP2(K0, K1)                                                                                         \
    asm volatile("s_nop 1\n\t"                                                                                       \
                 "v_add_u32_dpp v110, %[co], %[lb] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"              \
                 "v_add_u32_dpp v111, %[co], %[lb] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"              \
                 "s_nop 0\n\t"                                                                                       \
                 "ds_read_b128 v[112:115], v110\n\t"                                                                 \
                 "ds_read_b128 v[116:119], v110 offset:256\n\t"                                                      \
                 "ds_read_b128 v[120:123], v111\n\t"                                                                 \
                 "ds_read_b128 v[124:127], v111 offset:256\n\t"                                                      \
                 "s_waitcnt lgkmcnt(3)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c0], %[gv], v[112:113] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_fmac_f64_dpp %[c1], %[gv], v[114:115] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "s_waitcnt lgkmcnt(2)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c2], %[gv], v[116:117] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_fmac_f64_dpp %[c3], %[gv], v[118:119] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "s_waitcnt lgkmcnt(1)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c0], %[gv], v[120:121] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_fmac_f64_dpp %[c1], %[gv], v[122:123] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c2], %[gv], v[124:125] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_fmac_f64_dpp %[c3], %[gv], v[126:127] row_newbcast:" #K1 " row_mask:0xf bank_mask:0xf\n\t"       \
                 : [c0] "+v"(q0), [c1] "+v"(q1), [c2] "+v"(q2), [c3] "+v"(q3)                                        \
                 : [co] "v"(co), [lb] "v"(lb), [gv] "v"(gv)                                                          \
                 : "memory", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", \
                   "v121", "v122", "v123", "v124", "v125", "v126", "v127")

#define SBLAS_QSTEP1(K0)                                                                                             \
    asm volatile("s_nop 1\n\t"                                                                                       \
                 "v_add_u32_dpp v110, %[co], %[lb] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"              \
                 "s_nop 0\n\t"                                                                                       \
                 "ds_read_b128 v[112:115], v110\n\t"                                                                 \
                 "ds_read_b128 v[116:119], v110 offset:256\n\t"                                                      \
                 "s_waitcnt lgkmcnt(1)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c0], %[gv], v[112:113] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_fmac_f64_dpp %[c1], %[gv], v[114:115] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                          \
                 "v_fmac_f64_dpp %[c2], %[gv], v[116:117] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_fmac_f64_dpp %[c3], %[gv], v[118:119] row_newbcast:" #K0 " row_mask:0xf bank_mask:0xf\n\t"       \
                 : [c0] "+v"(q0), [c1] "+v"(q1), [c2] "+v"(q2), [c3] "+v"(q3)                                        \
                 : [co] "v"(co), [lb] "v"(lb), [gv] "v"(gv)                                                          \
                 : "memory", "v110", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119")

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

// steps come in pairs; a pair whose second step lies past the end runs it on masked slots (value 0, zero row)
#define SBLAS_QPAIR(K0, K1)                                                                                          \
    if (npairs > (K0 / 2)) {                                                                                         \
        SBLAS_QSTEP2(K0, K1);                                                                                        \
    }

// chunk of 64 nonzeros in quad order: lane 16q+k holds entry 4k+q
__device__ __forceinline__ void load_chunk_quad(const int *__restrict__ colidx, const double *__restrict__ val, int p,
                                                int pend, int eidx, int &c, double &v)
{
    const int idx = p + eidx;
    c = 0x7fffffff;
    v = 0.0;
    if (idx < pend) {
        c = colidx[idx];
        v = val[idx];
    }
}

template <int CH>
__global__ __launch_bounds__(1024) void spmm_window3_kernel(
    int rows, int cols, int npanels, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, const int2 *__restrict__ info, int ablate)
{
    constexpr int RPW = W3_RPW, R = W3_PANEL;
    static_assert(64 * (R + 1) <= 2 * W2_TILE, "C tile must fit in the (dead) B tile buffers");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *zero_row = smem + 2 * W2_TILE;
    int *sm_i = reinterpret_cast<int *>(smem + 2 * W2_TILE + 64); // [0] = bad

    const int panel = xcd_contiguous_panel(blockIdx.x, npanels);
    const int2 span = info[panel];
    if (span.x > span.y) return; // the direct kernel owns this panel

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    const int row0 = panel * R;
    const int col0 = blockIdx.y * 64;
    const unsigned ld32 = (unsigned)ldbt;
    const int t_lo = span.x / W2_ROWS, t_hi = span.y / W2_ROWS;
    const bool loader = wave >= W2_NCONS;
    const bool no_load = (ablate & 8) != 0, no_math = (ablate & 2) != 0; // diagnostics (wrong results)
    const bool no_bar = (ablate & 16) != 0, no_a = (ablate & 32) != 0, no_shift = (ablate & 64) != 0;

    if (tid < 64) zero_row[tid] = 0.0;
    if (tid == 0) sm_i[0] = 0;

    double acc[RPW][4];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = 0.0;
    double dacc[RPW]; // only used by the fallback (one column per lane)
#pragma unroll
    for (int r = 0; r < RPW; ++r) dacc[r] = 0.0;

    if (loader) {
        const int ltid = tid - W2_NCONS * 64;
        double2 st[16];
        auto fetch = [&](int t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int q = ltid + 256 * i;
                const int brow = t * W2_ROWS + (q >> 5);
                double2 x = make_double2(0.0, 0.0);
                if (brow < cols)
                    x = *reinterpret_cast<const double2 *>(Bt + (size_t)((unsigned)brow * ld32) + col0 + ((q & 31) << 1));
                st[i] = x;
            }
        };
        auto put = [&](int buf) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int q = ltid + 256 * i;
                *reinterpret_cast<double2 *>(smem + buf * W2_TILE + (q << 1)) = st[i];
            }
        };
        fetch(t_lo);
        put(0);
        if (t_lo < t_hi) fetch(t_lo + 1);
        __syncthreads(); // P
        for (int t = t_lo; t <= t_hi; ++t) {
            if (t < t_hi && !no_load) {
                put(((t - t_lo) & 1) ^ 1);
                if (t + 1 < t_hi) fetch(t + 2);
            }
            if (!no_bar) __syncthreads(); // E_t
        }
        __syncthreads(); // V
    } else {
        const int eidx = ((lane & 15) << 2) + (lane >> 4); // entry held by this lane inside a chunk / window
        int p0[RPW], len[RPW];
        int wc[RPW];           // current window: columns / values, quad order, left-aligned
        double wv[RPW];
        int sc[RPW][CH > 1 ? CH - 1 : 1];
        double sv[RPW][CH > 1 ? CH - 1 : 1];
        int wcnt[RPW], qcur[RPW], last_col[RPW];
        int bad = 0;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = row0 + wave * RPW + r;
            p0[r] = 0;
            len[r] = 0;
            if (row < rows) {
                p0[r] = wave_uniform(rowptr[row]);
                len[r] = wave_uniform(rowptr[row + 1]) - p0[r];
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            if (no_a) { // synthetic ascending columns inside the panel's span, no memory traffic
                wc[r] = span.x + eidx * 9;
                wv[r] = 1.0;
#pragma unroll
                for (int q = 1; q < CH; ++q) {
                    sc[r][q - 1] = min(span.x + (64 * q + eidx) * 9, span.y);
                    sv[r][q - 1] = 1.0;
                }
                continue;
            }
            load_chunk_quad(colidx, val, p0[r], p0[r] + len[r], eidx, wc[r], wv[r]);
#pragma unroll
            for (int q = 1; q < CH; ++q)
                load_chunk_quad(colidx, val, p0[r] + 64 * q, p0[r] + len[r], eidx, sc[r][q - 1], sv[r][q - 1]);
        }
        // ascending-column check of a chunk in quad order: the predecessor of entry e sits one DPP row up
        // (or, for the first DPP row, in the last DPP row one slot to the left)
        const int pred_lane = (lane >= 16) ? lane - 16 : lane + 47;
        auto chunk_ok = [&](int c, int cnt, int prev_last) -> bool {
            const int prev = __builtin_amdgcn_ds_bpermute(pred_lane << 2, c);
            return __ballot(eidx > 0 && eidx < cnt && c < prev) == 0ull && __builtin_amdgcn_readlane(c, 0) >= prev_last;
        };
        auto lane_of_entry = [](int e) { return ((e & 3) << 4) + (e >> 2); };
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            wcnt[r] = min(64, len[r]);
            qcur[r] = 0;
            last_col[r] = -1;
            if (wcnt[r] > 0) {
                if (!chunk_ok(wc[r], wcnt[r], -1)) bad = 1;
                last_col[r] = __builtin_amdgcn_readlane(wc[r], lane_of_entry(wcnt[r] - 1));
            }
        }
        __syncthreads(); // P
        for (int t = t_lo; t <= t_hi; ++t) {
            const int cur = (t - t_lo) & 1;
            const int tile_lo = t * W2_ROWS, tile_hi = tile_lo + W2_ROWS;
            const unsigned tile_base = (unsigned)(uintptr_t)(smem + cur * W2_TILE);
            const unsigned lb = tile_base + (unsigned)(lane & 15) * 16u;
            const unsigned zero_rel = (unsigned)(uintptr_t)zero_row - tile_base;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                for (;;) {
                    if (wcnt[r] == 0) {
                        // window empty: next stored chunk, if the row has one
                        if (64 * (qcur[r] + 1) >= len[r]) break;
                        qcur[r] += 1;
#pragma unroll
                        for (int q = 1; q < CH; ++q)
                            if (qcur[r] == q) {
                                wc[r] = sc[r][q - 1];
                                wv[r] = sv[r][q - 1];
                            }
                        wcnt[r] = min(64, len[r] - 64 * qcur[r]);
                        if (!chunk_ok(wc[r], wcnt[r], last_col[r])) bad = 1;
                        last_col[r] = __builtin_amdgcn_readlane(wc[r], lane_of_entry(wcnt[r] - 1));
                    }
                    // window is sorted and left-aligned: the entries of this tile are its first `take`
                    const int take = wave_uniform((int)__popcll(__ballot(eidx < wcnt[r] && wc[r] < tile_hi)));
                    if (take == 0) break;
                    {
                        const bool on = eidx < take;
                        const unsigned co = on ? ((unsigned)(wc[r] - tile_lo) << 9) : zero_rel;
                        const double gv = on ? wv[r] : 0.0;
                        const int npairs = (take + 7) >> 3; // pairs of 4-nonzero steps
                        double q0 = acc[r][0], q1 = acc[r][1], q2 = acc[r][2], q3 = acc[r][3];
                        if (!no_math) {
                        SBLAS_QPAIR(0, 1)
                        SBLAS_QPAIR(2, 3)
                        SBLAS_QPAIR(4, 5)
                        SBLAS_QPAIR(6, 7)
                        SBLAS_QPAIR(8, 9)
                        SBLAS_QPAIR(10, 11)
                        SBLAS_QPAIR(12, 13)
                        SBLAS_QPAIR(14, 15)
                        }
                        acc[r][0] = q0;
                        acc[r][1] = q1;
                        acc[r][2] = q2;
                        acc[r][3] = q3;
                    }
                    // slide the window: entry e of the new window is entry e+take of the old one
                    const int rest = wcnt[r] - take;
                    if (rest > 0 && !no_shift) {
                        const int se = eidx + take;
                        const int src = lane_of_entry(se & 63) << 2;
                        // lanes >= rest receive stale entries; every use of the window is masked by wcnt, so they
                        // are never looked at -- and with no select here the shuffle is not waited for until the
                        // next visit of this row
                        const int nc = __builtin_amdgcn_ds_bpermute(src, wc[r]);
                        const int nlo = __builtin_amdgcn_ds_bpermute(src, __double2loint(wv[r]));
                        const int nhi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(wv[r]));
                        wc[r] = nc;
                        wv[r] = __hiloint2double(nhi, nlo);
                    }
                    wcnt[r] = rest;
                    if (rest > 0) break; // what is left belongs to later tiles
                }
            }
            if (!no_bar) __syncthreads(); // E_t
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
            if (wcnt[r] != 0 || 64 * (qcur[r] + 1) < len[r]) bad = 1; // unconsumed nonzeros
        if (bad && lane == 0) atomicOr(&sm_i[0], 1);
        __syncthreads(); // V
        if (sm_i[0] != 0) {
            const unsigned lane_off = (unsigned)(col0 + lane);
#pragma unroll
            for (int r = 0; r < RPW; ++r)
                dacc[r] = row_direct(colidx, val, Bt, ld32, lane_off, lane, p0[r], p0[r] + len[r]);
        } else {
            // fold the four DPP rows (partial sums over different nonzeros of the same row)
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double x = acc[r][j];
                    x += __shfl_xor(x, 16, WAVE);
                    x += __shfl_xor(x, 32, WAVE);
                    acc[r][j] = x;
                }
        }
    }
    const bool fell_back = sm_i[0] != 0;
    if (tid == 0 && blockIdx.y == 0) atomicAdd(&g_panel_stats[fell_back ? 2 : 0], 1ull);

    // park the panel as [column][row] in the (now dead) tile buffers and write it back along rows
    double *ctile = smem;
    if (!loader) {
        if (fell_back) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) ctile[lane * (R + 1) + wave * RPW + r] = dacc[r];
        } else if (lane < 16) {
            const int jj = lane;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int rr = wave * RPW + r;
                ctile[(2 * jj) * (R + 1) + rr] = acc[r][0];
                ctile[(2 * jj + 1) * (R + 1) + rr] = acc[r][1];
                ctile[(32 + 2 * jj) * (R + 1) + rr] = acc[r][2];
                ctile[(33 + 2 * jj) * (R + 1) + rr] = acc[r][3];
            }
        }
    }
    __syncthreads(); // F
    const int nrows = min(R, rows - row0);
    const int ncols = min(64, n - col0);
    for (int idx = tid; idx < 64 * R; idx += 1024) {
        const int r = idx % R, j = idx / R;
        if (r < nrows && j < ncols) {
            double *dst = C + (int64_t)(col0 + j) * ldc + (row0 + r);
            const double sres = alpha * ctile[j * (R + 1) + r];
            *dst = (beta == 0.0) ? sres : fma(beta, *dst, sres);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 2, windowed form, fourth generation: streaming quad consumer.
//
// Ablation of the third generation: > half of its time is fixed cost per panel (A burst, per-visit bookkeeping,
// barriers, epilogue) because a wave can keep only three whole rows in registers -> 36-row panels.  Here a
// consumer wave keeps NO row data resident: at every (row, tile) visit it uses a 64-entry window loaded from
// col_idx/val at the row's cursor (quad order, so the first `take` entries are DPP steps 0..take/4 -- no shift, no
// chunk switch, no limit on the row length), issued NBUF-1 visits ahead into a ring of NBUF register buffers with
// hand-counted vmcnt waits (hipcc's own bookkeeping turns conservative in this loop and would wait for the prefetch
// just issued).  The re-reads hit L2 (the window advances ~12 entries per tile).  State per row = cursor + 4
// accumulators.  Measured: what matters is the latency of a visit, not the panel size -- 4 rows per wave with a
// 4-deep ring runs 509 us, 6 rows / 3-deep 842 us, 8 rows / 4-deep 1244 us, 2 rows / 2-deep 653 us.
// Every consumed entry is checked to lie inside the current tile and the consumed set to be a prefix of the window;
// anything else marks the panel for the direct-loop fallback.
// ---------------------------------------------------------------------------------------------
constexpr int W4_RPW = 4;
constexpr int W4_PANEL = W2_NCONS * W4_RPW; // rows per panel
constexpr int W4_NBUF = 4;                  // window buffers: the window of a visit is issued NBUF-1 visits ahead
constexpr int W4_PF_DIST = 32;              // A prefetch: first entry touched, counted from the row's cursor

// Window loads of the streaming consumer.  They are buffer loads through two structured descriptors (col_idx:
// stride 4, val: stride 8) that start at the first nonzero of the wave's rows: the per-lane operand is just the
// entry number (cursor + lane's window slot, ONE vector add per window), entries past the end of the arrays read 0
// instead of faulting, and no address arithmetic is left in the visit loop.  The loads are hidden from hipcc's vmcnt
// bookkeeping (which turns conservative in the visit loop and would wait for the prefetch just issued): issued in
// one asm statement and retired by a counted wait that names their destinations, so no compiler-generated use can
// be scheduled before the data has landed (cdna_hip_programming.md section 5.7, form (ii)).  Entries past the row end
// are ignored by the caller (every use is masked by the window count).
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
__device__ __forceinline__ void window_issue(sblas_rsrc_t rc, sblas_rsrc_t rv, int cur, int eidx, int &c, double &v)
{
    const int idx = cur + eidx;
    asm volatile("buffer_load_dword %0, %2, %3, 0 idxen\n\tbuffer_load_dwordx2 %1, %2, %4, 0 idxen"
                 : "=&v"(c), "=&v"(v)
                 : "v"(idx), "s"(rc), "s"(rv)
                 : "memory");
}
// Which window entries belong to the tile [tile_lo, tile_lo + 128)?  For those: LDS byte offset of their B row and
// their value; for all others the offset of the all-zero row and value 0.  `m` = lane mask of the entries taken.
// Seven vector instructions, written out because the compiler's version needs fourteen.
__device__ __forceinline__ void window_select(int wc, double wv, int tile_lo, int cnt, int eidx, unsigned zero_rel,
                                              unsigned &co, double &gv, unsigned long long &m)
{
    int glo, ghi;
    asm volatile("v_subrev_u32 %[co], %[tlo], %[wc]\n\t"
                 "v_cmp_gt_i32 %[m], %[cnt], %[eidx]\n\t"
                 "v_cmp_gt_u32 vcc, 0x80, %[co]\n\t"
                 "v_lshlrev_b32 %[co], 9, %[co]\n\t"
                 "s_and_b64 vcc, vcc, %[m]\n\t"
                 "s_mov_b64 %[m], vcc\n\t"
                 "v_cndmask_b32 %[co], %[zr], %[co], vcc\n\t"
                 "v_cndmask_b32 %[glo], 0, %[vlo], vcc\n\t"
                 "v_cndmask_b32 %[ghi], 0, %[vhi], vcc"
                 : [co] "=&v"(co), [m] "=&s"(m), [glo] "=&v"(glo), [ghi] "=&v"(ghi)
                 : [tlo] "s"(tile_lo), [wc] "v"(wc), [cnt] "s"(cnt), [eidx] "v"(eidx), [zr] "v"(zero_rel),
                   [vlo] "v"(__double2loint(wv)), [vhi] "v"(__double2hiint(wv))
                 : "vcc", "scc");
    gv = __hiloint2double(ghi, glo);
}
__device__ __forceinline__ int mask_count(unsigned long long m)
{
    int n;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(m) : "scc");
    return n;
}
// wait until at most `NEWER` younger vector-memory operations are outstanding
template <int NEWER> __device__ __forceinline__ void window_wait(int &c, double &v)
{
    static_assert(NEWER == 0 || NEWER == 2 || NEWER == 4 || NEWER == 6, "two loads per window");
    if (NEWER == 6) asm volatile("s_waitcnt vmcnt(6)" : "+v"(c), "+v"(v)::"memory");
    else if (NEWER == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(c), "+v"(v)::"memory");
    else if (NEWER == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(c), "+v"(v)::"memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(c), "+v"(v)::"memory");
}

// SBLAS_ABLATE bits understood by this kernel (diagnostics; every one of them produces wrong results):
//   0x10000 no LDS reads / FMAs      0x20000 no tile DMA      0x40000 no per-tile barrier     0x80000 no A prefetch
//   0x10000000 windows used without waiting (and no fallback)   0x20000000 no window loads in the tile loop
//   0x40000000 no tile loop at all   bits 8..15: pace the DMA (64-cycle sleeps)   bits 20..27: A prefetch distance
template <bool ABL> // ABL: the diagnostic switches are compiled in (launched only when SBLAS_ABLATE is set)
__global__ __launch_bounds__(1024) void spmm_window4_kernel(
    int rows, int cols, int npanels, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, const int2 *__restrict__ info, int ablate, int nnz)
{
    constexpr int RPW = W4_RPW, R = W4_PANEL;
    constexpr int WIN = 32; // window entries fetched per visit (the L2 must keep every row's window between two visits)
    static_assert(64 * (R + 1) <= 2 * W2_TILE, "C tile must fit in the (dead) B tile buffers");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *zero_row = smem + 2 * W2_TILE;
    int *sm_i = reinterpret_cast<int *>(smem + 2 * W2_TILE + 64); // [0] = bad, [8..8+R) = published row cursors

    const int panel = xcd_contiguous_panel(blockIdx.x, npanels);
    const int2 span = info[panel];
    if (span.x > span.y) return; // the direct kernel owns this panel

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    const int row0 = panel * R;
    const int col0 = blockIdx.y * 64;
    const unsigned ld32 = (unsigned)ldbt;
    const int t_lo = span.x / W2_ROWS, t_hi = (ABL && (ablate & 0x40000000)) ? t_lo - 1 : span.y / W2_ROWS;
    const bool loader = wave >= W2_NCONS;
    const bool no_bar = ABL && (ablate & 0x40000) != 0;

    if (tid < 64) zero_row[tid] = 0.0;
    if (tid == 0) sm_i[0] = 0;

    double acc[RPW][4];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = 0.0;

    if (loader) {
        // ---------------- loader waves: LDS-DMA, no staging registers ----------------
        // tools/tile_load_bench.hip: four waves staging through registers move 14 B/clk/CU from L2, four waves issuing
        // global_load_lds_dwordx4 move 40 B/clk/CU.  Piece q (16 bytes) of a tile lives at LDS byte q*16 and comes from
        // Bt row t*128 + q/32, bytes (q%32)*16; a wave-level DMA writes 64 consecutive pieces (wave-uniform LDS base +
        // lane*16), so loader wave w issues pieces 64*w + 256*i + lane, i = 0..15.  Rows past the end of B are clamped
        // to row `cols`, the all-zero row of the workspace.
        const int lw = wave - W2_NCONS;
        const unsigned piece_off = (unsigned)(col0 + ((lane & 31) << 1)) * 8u; // bytes inside a Bt row
        const unsigned ldb8 = ld32 * 8u;
        const char *bt_bytes = reinterpret_cast<const char *>(Bt);
        const int pace = ABL ? (ablate >> 8) & 0xff : 0;
        const bool no_dma = ABL && (ablate & 0x20000) != 0;
        auto dma_tile = [&](int t, int buf) {
            const int r_first = t * W2_ROWS + lw * 2 + (lane >> 5);
            char *lds_wave = reinterpret_cast<char *>(smem + buf * W2_TILE) + lw * 1024;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned brow = (unsigned)min(r_first + 8 * i, cols);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(bt_bytes + (size_t)(brow * ldb8 + piece_off)),
                    (__attribute__((address_space(3))) void *)(lds_wave + i * 4096), 16, 0, 0);
                if ((i & 1) == 1)
                    for (int z = 0; z < pace; ++z) __builtin_amdgcn_s_sleep(1);
            }
        };
        // A prefetch: the loader waves touch the col_idx/val cache lines a little ahead of every row's cursor
        // (published by the consumers once per tile) so that the consumers' window loads find them in L2: five lanes
        // per row -- two col_idx lines, three val lines, 48 entries starting pf_dist entries past the cursor.  The
        // touch is issued after the tile DMA and never waited for (counted vmcnt), so HBM latency stays off the
        // barrier path.  Worth 3 % on the bench matrix.
        static_assert(R == 4 * 12, "prefetch lanes: 12 rows per loader wave, 5 lanes per row");
        const int pf_dist = (ABL && ((ablate >> 20) & 0xff)) ? ((ablate >> 20) & 0xff) : W4_PF_DIST;
        const bool no_pf = ABL && (ablate & 0x80000) != 0;
        const int prow = min(lw * 12 + lane / 5, R - 1);
        const int pk = lane % 5;
        const int p_off = pf_dist + (pk < 2 ? 32 * pk : 16 * (pk - 2));
        const int p_last = max(rowptr[min(row0 + prow, rows - 1) + 1] - 1, 0);
        const char *p_base = pk < 2 ? reinterpret_cast<const char *>(colidx) : reinterpret_cast<const char *>(val);
        const int p_shift = pk < 2 ? 2 : 3;
        int pf_sink = 0; // landing register of the touches: read-write in every statement below so that the register
                         // allocator never lends it to another value while a touch is still in flight
        auto touch = [&](int cursor) {
            const int idx = min(cursor + p_off, p_last);
            const char *p = p_base + ((size_t)(unsigned)idx << p_shift);
            asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(p) : "memory");
        };
        dma_tile(t_lo, 0);
        __syncthreads(); // P (the barrier drains the DMA: vmcnt(0)); the consumers have published the row starts
        for (int t = t_lo; t <= t_hi; ++t) {
            const int cursor = sm_i[8 + prow]; // as of the end of the previous tile
            if (t < t_hi && !no_dma) dma_tile(t + 1, ((t - t_lo) & 1) ^ 1); // that buffer was last read before the previous barrier
            if (!no_pf) {
                touch(cursor);
                // everything but the touch just issued has landed: the tile DMA and the previous tile's touch
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            }
            if (!no_bar) {
                if (no_pf) __syncthreads(); // E_t
                else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // E_t, without draining the touch
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink)::"memory");
        __syncthreads(); // V
    } else {
        const int eidx = ((lane & 15) << 2) + (lane >> 4); // window entry held by this lane (quad order)
        // entry number used for the loads: the lanes past the window size get an index that no descriptor covers, so
        // they cost no memory traffic (and read 0)
        const int eload = eidx < WIN ? eidx : 0x40000000;
        const bool no_math = ABL && (ablate & 0x10000) != 0, no_wait = ABL && (ablate & 0x10000000) != 0;
        const bool no_issue = ABL && (ablate & 0x20000000) != 0;
        // cursors are kept relative to the first nonzero of the wave's rows (base of the two descriptors)
        const int wrow = min(row0 + wave * RPW, rows);
        const int wstart = wave_uniform(rowptr[wrow]);
        const sblas_rsrc_t rc = make_rsrc(colidx + wstart, 4u, (unsigned)(nnz - wstart));
        const sblas_rsrc_t rv = make_rsrc(val + wstart, 8u, (unsigned)(nnz - wstart));
        int cur[RPW], end[RPW];
        int bad = 0;
        unsigned long long viol = 0ull; // lanes whose entry broke the "consumed set = window prefix" expectation
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = row0 + wave * RPW + r;
            cur[r] = end[r] = 0;
            if (row < rows) {
                cur[r] = wave_uniform(rowptr[row]) - wstart;
                end[r] = wave_uniform(rowptr[row + 1]) - wstart;
            }
        }
        // NBUF window buffers used round-robin by the rows of this wave (RPW is a multiple of NBUF, so the buffer of a
        // row is the same in every tile); the window of a visit is issued NBUF-1 visits ahead.
        static_assert(RPW % W4_NBUF == 0 && W4_NBUF >= 2 && W4_NBUF <= 4, "buffer ring must divide the rows of a wave");
        constexpr int NB = W4_NBUF, AHEAD = W4_NBUF - 1;
        int wcb[NB];
        double wvb[NB];
#pragma unroll
        for (int r = 0; r < AHEAD; ++r) window_issue(rc, rv, cur[r], eload, wcb[r], wvb[r]);
        static_assert(RPW == 4, "the cursors of a wave are published as one int4");
        int4 *cur_pub = reinterpret_cast<int4 *>(sm_i + 8) + wave; // read by the loader waves' prefetch (absolute)
        if (lane == 0) *cur_pub = make_int4(cur[0] + wstart, cur[1] + wstart, cur[2] + wstart, cur[3] + wstart);
        __syncthreads(); // P
        for (int t = t_lo; t <= t_hi; ++t) {
            const int cb = (t - t_lo) & 1;
            const int tile_lo = t * W2_ROWS;
            const unsigned tile_base = (unsigned)(uintptr_t)(smem + cb * W2_TILE);
            const unsigned lb = tile_base + (unsigned)(lane & 15) * 16u;
            const unsigned zero_rel = (unsigned)(uintptr_t)zero_row - tile_base;
            // one visit per row, written as a generic lambda over a compile-time row index so that every register
            // array index below is a constant (a plain unrolled loop was left rolled by the optimiser)
            auto visit = [&](auto rc_) {
                constexpr int r = decltype(rc_)::value;
                // issue the window of the visit AHEAD positions later (wrapping into the next tile: that row's cursor
                // was already advanced in this tile)
                constexpr int rn = (r + AHEAD) % RPW;
                if (!no_issue) window_issue(rc, rv, cur[rn], eload, wcb[rn % NB], wvb[rn % NB]);
                // this row's window was issued AHEAD visits ago: only the AHEAD younger windows may still be in flight
                if (!no_wait) window_wait<2 * AHEAD>(wcb[r % NB], wvb[r % NB]);
                double &q0 = acc[r][0], &q1 = acc[r][1], &q2 = acc[r][2], &q3 = acc[r][3];
                // The lanes decide for themselves which window entries belong to this tile, and the first sixteen slots
                // are processed straight away (masked slots: value 0, zero row) -- no scalar round trip (ballot ->
                // popcount -> branch) sits in front of the LDS reads, and a row that is already finished (count <= 0)
                // simply runs on masked slots.  Count and validation follow.  One loop body serves the rare cases too
                // (a single back edge and a single exit, so the accumulators stay in place): more than 16 entries in
                // this tile -> drop the 16 just done by shifting the window four lanes down every DPP row and go again;
                // whole window consumed -> fetch the next one and go again.
                static_assert(W2_ROWS == 128, "window_select compares against a 128-row tile");
                int cnt = min(WIN, end[r] - cur[r]);
                int wc = wcb[r % NB];
                double wv = wvb[r % NB];
                for (;;) {
                    unsigned co;
                    double gv;
                    unsigned long long m;
                    window_select(wc, wv, tile_lo, cnt, eidx, zero_rel, co, gv, m);
                    if (!no_math) { SBLAS_QSTEP4(0, 1, 2, 3); }
                    const int take = mask_count(m);
                    // with ascending columns the entries of this tile are exactly the first `take` of the window; any
                    // other pattern is remembered and the panel recomputed at the end (the loop itself stays safe:
                    // masked slots read the zero row, the cursor never passes the row end)
                    viol |= m ^ __builtin_amdgcn_ballot_w64(eidx < take);
                    const bool more = take > 16 || (take >= cnt && cur[r] + take < end[r]);
                    if (__builtin_expect(!more, 1)) {
                        cur[r] += take;
                        break;
                    }
                    if (take > 16) {
                        cur[r] += 16;
                        cnt -= 16;
                        int lo = __double2loint(wv), hi = __double2hiint(wv);
                        asm volatile("s_nop 1\n\t"
                                     "v_mov_b32_dpp %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                                     "v_mov_b32_dpp %1, %1 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                                     "v_mov_b32_dpp %2, %2 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                                     : "+v"(wc), "+v"(lo), "+v"(hi));
                        wv = __hiloint2double(hi, lo);
                    } else {
                        cur[r] += take;
                        cnt = min(WIN, end[r] - cur[r]);
                        window_issue(rc, rv, cur[r], eload, wc, wv);
                        window_wait<0>(wc, wv);
                    }
                }
            };
            visit(std::integral_constant<int, 0>{});
            visit(std::integral_constant<int, 1>{});
            visit(std::integral_constant<int, 2>{});
            visit(std::integral_constant<int, 3>{});
            if (lane == 0) *cur_pub = make_int4(cur[0] + wstart, cur[1] + wstart, cur[2] + wstart, cur[3] + wstart);
            if (!no_bar) __syncthreads(); // E_t
        }
        window_wait<0>(wcb[0], wvb[0]); // retire the last (unused) prefetches before the registers are reused
        if (viol != 0ull) bad = 1;
#pragma unroll
        for (int r = 0; r < RPW; ++r)
            if (cur[r] < end[r]) bad = 1; // unconsumed nonzeros
        if (no_wait) bad = 0;
        if (bad && lane == 0) atomicOr(&sm_i[0], 1);
        __syncthreads(); // V
        if (sm_i[0] != 0) {
            // recompute straight from L2, one column per lane, and store in the quad accumulator layout's slot 0
            const unsigned lane_off = (unsigned)(col0 + lane);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = row0 + wave * RPW + r;
                int a = 0, b = 0;
                if (row < rows) {
                    a = wave_uniform(rowptr[row]);
                    b = wave_uniform(rowptr[row + 1]);
                }
                acc[r][0] = row_direct(colidx, val, Bt, ld32, lane_off, lane, a, b);
            }
        } else {
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double x = acc[r][j];
                    x += __shfl_xor(x, 16, WAVE);
                    x += __shfl_xor(x, 32, WAVE);
                    acc[r][j] = x;
                }
        }
    }
    const bool fell_back = sm_i[0] != 0;
    if (tid == 0 && blockIdx.y == 0) atomicAdd(&g_panel_stats[fell_back ? 2 : 0], 1ull);

    double *ctile = smem;
    if (!loader) {
        if (fell_back) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) ctile[lane * (R + 1) + wave * RPW + r] = acc[r][0];
        } else if (lane < 16) {
            const int jj = lane;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int rr = wave * RPW + r;
                ctile[(2 * jj) * (R + 1) + rr] = acc[r][0];
                ctile[(2 * jj + 1) * (R + 1) + rr] = acc[r][1];
                ctile[(32 + 2 * jj) * (R + 1) + rr] = acc[r][2];
                ctile[(33 + 2 * jj) * (R + 1) + rr] = acc[r][3];
            }
        }
    }
    __syncthreads(); // F
    const int nrows = min(R, rows - row0);
    const int ncols = min(64, n - col0);
    for (int idx = tid; idx < 64 * R; idx += 1024) {
        const int r = idx % R, j = idx / R;
        if (r < nrows && j < ncols) {
            double *dst = C + (int64_t)(col0 + j) * ldc + (row0 + r);
            const double sres = alpha * ctile[j * (R + 1) + r];
            *dst = (beta == 0.0) ? sres : fma(beta, *dst, sres);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 2, windowed form, fifth generation: no loader waves.
//
// SQ counters on the fourth generation: vector ALU 38 % busy, LDS 37 %, and a consumer wave spends ~1100 cycles on
// a visit whose instructions would fit in ~450 -- a wave issues at most one instruction every four cycles and the
// visit is one serial chain, so throughput is (waves that consume) / (visit latency), and a quarter of the
// waves were loaders that sleep at the barrier.  Here all 16 waves consume (64-row panels: 25 % less tile traffic
// per nonzero as well) and every wave issues its own 4 KB share of the next tile's LDS-DMA at the top of a tile.
// The DMA shares the in-order vmcnt queue with the window loads, so the counted waits grow by the four DMA
// operations (10 instead of 6) and the wait of a tile's last visit (everything older than the windows of the
// last three visits has landed) also covers the DMA -- no extra wait in front of the barrier.  The same number of
// vector-memory operations is issued in every tile (the last tile fetches a tile nobody reads), which keeps the
// counts exact.
// ---------------------------------------------------------------------------------------------
constexpr int W5_RPW = 4;
constexpr int W5_PANEL = 16 * W5_RPW; // 64 rows

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
// One vector-memory instruction per window (the address unit is the busiest resource of this kernel: a wave64 load
// occupies it for 16 cycles whatever it fetches).  Lanes k = 0..7 of every DPP row fetch the VALUE of window entry
// 4k + q (q = DPP row), lanes 8..15 fetch eight bytes of col_idx starting at the same entry (the first dword is
// used): the broadcast of step k then takes the Bt row offset from lane k + 8 and the value from lane k.
__device__ __forceinline__ void window_issue5(const char *addr, double &d)
{
    asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(d) : "v"(addr) : "memory");
}
template <int NEWER> __device__ __forceinline__ void window_wait5(double &d)
{
    static_assert(NEWER == 0 || NEWER == 3 || NEWER == 7, "counts of the fifth-generation visit loop");
    if (NEWER == 7) asm volatile("s_waitcnt vmcnt(7)" : "+v"(d)::"memory");
    else if (NEWER == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(d)::"memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(d)::"memory");
}
constexpr unsigned long long W5_COL_LANES = 0xff00ff00ff00ff00ull, W5_VAL_LANES = 0x00ff00ff00ff00ffull;
// tile membership of the window entries: decided on the col_idx lanes, copied eight lanes down for the value lanes.
// co (col_idx lanes) = LDS offset of the B row or of the zero row; gv (value lanes) = value or 0; m = value lanes taken
__device__ __forceinline__ void window_select5(double d, int tile_lo, int cnt, int eidx, unsigned zero_rel,
                                               unsigned &co, double &gv, unsigned long long &m)
{
    int glo, ghi;
    asm volatile("v_subrev_u32 %[co], %[tlo], %[d0]\n\t"
                 "v_cmp_gt_i32 %[m], %[cnt], %[eidx]\n\t"
                 "v_cmp_gt_u32 vcc, 0x80, %[co]\n\t"
                 "v_lshlrev_b32 %[co], 9, %[co]\n\t"
                 "s_and_b64 vcc, vcc, %[m]\n\t"
                 "s_and_b64 vcc, vcc, %[cl]\n\t"
                 "s_lshr_b64 %[m], vcc, 8\n\t"
                 "s_or_b64 vcc, vcc, %[m]\n\t"
                 "v_cndmask_b32 %[co], %[zr], %[co], vcc\n\t"
                 "v_cndmask_b32 %[glo], 0, %[d0], vcc\n\t"
                 "v_cndmask_b32 %[ghi], 0, %[d1], vcc"
                 : [co] "=&v"(co), [m] "=&s"(m), [glo] "=&v"(glo), [ghi] "=&v"(ghi)
                 : [tlo] "s"(tile_lo), [cnt] "s"(cnt), [eidx] "v"(eidx), [zr] "v"(zero_rel), [cl] "s"(W5_COL_LANES),
                   [d0] "v"(__double2loint(d)), [d1] "v"(__double2hiint(d))
                 : "vcc", "scc");
    gv = __hiloint2double(ghi, glo);
}
// sixteen slots: steps K0..K3, Bt row offset from lane K + 8, value from lane K of every DPP row
#define SBLAS_QSTEP4M(K0, K1, K2, K3, C0, C1, C2, C3)                                                                \
    asm volatile("s_nop 1\n\t"                                                                                       \
                 "v_add_u32_dpp v92, %[co], %[lb] row_newbcast:" #C0 " row_mask:0xf bank_mask:0xf\n\t"               \
                 "v_add_u32_dpp v93, %[co], %[lb] row_newbcast:" #C1 " row_mask:0xf bank_mask:0xf\n\t"               \
                 "v_add_u32_dpp v94, %[co], %[lb] row_newbcast:" #C2 " row_mask:0xf bank_mask:0xf\n\t"               \
                 "v_add_u32_dpp v95, %[co], %[lb] row_newbcast:" #C3 " row_mask:0xf bank_mask:0xf\n\t"               \
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

// SBLAS_ABLATE bits understood (diagnostics, wrong results): 0x10000 no LDS reads / FMAs, 0x40000 no per-tile
// barrier, 0x40000000 no tile loop at all.
template <bool ABL>
__global__ __launch_bounds__(1024) void spmm_window5_kernel(
    int rows, int cols, int npanels, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, const int2 *__restrict__ info, int ablate, int nnz)
{
    constexpr int RPW = W5_RPW, R = W5_PANEL;
    constexpr int WIN = 32; // window entries fetched per visit
    static_assert(64 * (R + 1) <= 2 * W2_TILE, "C tile must fit in the (dead) B tile buffers");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *zero_row = smem + 2 * W2_TILE;
    int *sm_i = reinterpret_cast<int *>(smem + 2 * W2_TILE + 64); // [0] = bad

    const int panel = xcd_contiguous_panel(blockIdx.x, npanels);
    const int2 span = info[panel];
    if (span.x > span.y) return; // the direct kernel owns this panel

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    const int row0 = panel * R;
    const int col0 = blockIdx.y * 64;
    const unsigned ld32 = (unsigned)ldbt;
    const int t_lo = span.x / W2_ROWS, t_hi = (ABL && (ablate & 0x40000000)) ? t_lo - 1 : span.y / W2_ROWS;
    const bool no_bar = ABL && (ablate & 0x40000) != 0, no_math = ABL && (ablate & 0x10000) != 0;

    if (tid < 64) zero_row[tid] = 0.0;
    if (tid == 0) sm_i[0] = 0;

    double acc[RPW][4];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = 0.0;

    // ---- tile DMA: wave w moves Bt rows 8w..8w+7 of a tile (4 instructions of two rows each: lanes 0-31 the
    // first, lanes 32-63 the second); tile row j lives at LDS byte j*512.  Rows past the end of B are clamped to row
    // `cols`, the all-zero row of the workspace: the common case (all eight rows exist) runs on scalar addresses.
    const unsigned ldb8 = ld32 * 8u;
    const unsigned piece_off = (unsigned)(col0 + ((lane & 31) << 1)) * 8u; // bytes inside a Bt row
    const unsigned pair_off = (unsigned)(lane >> 5) * ldb8 + piece_off;
    const char *bt_bytes = reinterpret_cast<const char *>(Bt);
    auto dma_tile = [&](int t, int buf) {
        const int r0 = t * W2_ROWS + wave * 8;
        const unsigned lds0 = (unsigned)(uintptr_t)(smem + buf * W2_TILE) + (unsigned)wave * 4096u;
        if (r0 + 7 <= cols) {
            const char *p = bt_bytes + (size_t)((unsigned)r0 * ldb8);
#pragma unroll
            for (int i = 0; i < 4; ++i) dma_rows_scalar(lds0 + i * 1024u, pair_off, p + (size_t)(2u * i) * ldb8);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned brow = (unsigned)min(r0 + 2 * i + (lane >> 5), cols);
                dma_rows_vector(lds0 + i * 1024u, bt_bytes + (size_t)(brow * ldb8 + piece_off));
            }
        }
    };

    // window entry held by this lane (quad order; value in lanes 0-7 of a DPP row, col_idx in lanes 8-15)
    const int eidx = ((lane & 7) << 2) + (lane >> 4);
    const bool col_lane = (lane & 8) != 0;
    // cursors are kept relative to the first nonzero of the wave's rows
    const int wrow = min(row0 + wave * RPW, rows);
    const int wstart = wave_uniform(rowptr[wrow]);
    // loads are clamped to the arrays (clamped entries lie past `cnt`).  A col_idx lane fetches eight bytes, so it is
    // clamped one entry lower: the very last nonzero of the matrix would then read its neighbour's column -- the
    // classifier hands the panel that contains it to the direct kernel (`exclude_tail`).
    const int last_rel = max(nnz - (col_lane ? 2 : 1) - wstart, 0);
    const char *lane_base = col_lane ? reinterpret_cast<const char *>(colidx + wstart)
                                     : reinterpret_cast<const char *>(val + wstart);
    const int lane_shift = col_lane ? 2 : 3;
    auto win_addr = [&](int cursor) {
        const int idx = min(cursor + eidx, last_rel);
        return lane_base + ((size_t)(unsigned)idx << lane_shift);
    };
    int cur[RPW], end[RPW];
    int bad = 0;
    unsigned long long viol = 0ull; // lanes whose entry broke the "consumed set = window prefix" expectation
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + wave * RPW + r;
        cur[r] = end[r] = 0;
        if (row < rows) {
            cur[r] = wave_uniform(rowptr[row]) - wstart;
            end[r] = wave_uniform(rowptr[row + 1]) - wstart;
        }
    }
    constexpr int NB = 4, AHEAD = 3;
    static_assert(RPW == NB, "one window buffer per row of the wave");
    double wdb[NB];
    dma_tile(t_lo, 0);
#pragma unroll
    for (int r = 0; r < AHEAD; ++r) window_issue5(win_addr(cur[r]), wdb[r]);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); // the first tile has landed (the three windows may be in flight)
    __syncthreads(); // P
    for (int t = t_lo; t <= t_hi; ++t) {
        const int cb = (t - t_lo) & 1;
        dma_tile(t + 1, cb ^ 1); // that buffer was last read before the previous barrier; past t_hi: a tile nobody reads
        const int tile_lo = t * W2_ROWS;
        const unsigned tile_base = (unsigned)(uintptr_t)(smem + cb * W2_TILE);
        const unsigned lb = tile_base + (unsigned)(lane & 15) * 16u;
        const unsigned zero_rel = (unsigned)(uintptr_t)zero_row - tile_base;
        auto visit = [&](auto rc_) {
            constexpr int r = decltype(rc_)::value;
            constexpr int rn = (r + AHEAD) % RPW;
            window_issue5(win_addr(cur[rn]), wdb[rn]);
            // operations younger than this row's window: the windows of the two visits in between, this tile's four
            // DMA operations unless they are older (last visit of the tile), and the window just issued
            window_wait5<(r == RPW - 1) ? 3 : 7>(wdb[r]);
            double &q0 = acc[r][0], &q1 = acc[r][1], &q2 = acc[r][2], &q3 = acc[r][3];
            static_assert(W2_ROWS == 128, "window_select compares against a 128-row tile");
            int cnt = min(WIN, end[r] - cur[r]);
            double wd = wdb[r];
            for (;;) { // see the fourth generation for the structure of this loop
                unsigned co;
                double gv;
                unsigned long long m;
                window_select5(wd, tile_lo, cnt, eidx, zero_rel, co, gv, m);
                if (!no_math) { SBLAS_QSTEP4M(0, 1, 2, 3, 8, 9, 10, 11); }
                const int take = mask_count(m);
                viol |= m ^ (__builtin_amdgcn_ballot_w64(eidx < take) & W5_VAL_LANES);
                const bool more = take > 16 || (take >= cnt && cur[r] + take < end[r]);
                if (__builtin_expect(!more, 1)) {
                    cur[r] += take;
                    break;
                }
                if (take > 16) {
                    cur[r] += 16;
                    cnt -= 16;
                    // entries 16.. move to 0..: four lanes down, in the value half and in the col_idx half of every DPP row
                    int lo = __double2loint(wd), hi = __double2hiint(wd);
                    asm volatile("s_nop 1\n\t"
                                 "v_mov_b32_dpp %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                                 "v_mov_b32_dpp %1, %1 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                                 : "+v"(lo), "+v"(hi));
                    wd = __hiloint2double(hi, lo);
                } else {
                    cur[r] += take;
                    cnt = min(WIN, end[r] - cur[r]);
                    window_issue5(win_addr(cur[r]), wd);
                    window_wait5<0>(wd); // drains the queue: the counted waits that follow stay correct (no-ops)
                }
            }
        };
        visit(std::integral_constant<int, 0>{});
        visit(std::integral_constant<int, 1>{});
        visit(std::integral_constant<int, 2>{});
        visit(std::integral_constant<int, 3>{});
        if (!no_bar) __syncthreads(); // E_t
    }
    window_wait5<0>(wdb[0]); // retire the last (unused) window and tile fetches
    if (viol != 0ull) bad = 1;
#pragma unroll
    for (int r = 0; r < RPW; ++r)
        if (cur[r] < end[r]) bad = 1; // unconsumed nonzeros
    if (ABL && ablate != 0) bad = 0;
    if (bad && lane == 0) atomicOr(&sm_i[0], 1);
    __syncthreads(); // V
    const bool fell_back = sm_i[0] != 0;
    if (fell_back) {
        // recompute straight from L2, one column per lane, and store in the quad accumulator layout's slot 0
        const unsigned lane_off = (unsigned)(col0 + lane);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = row0 + wave * RPW + r;
            int a = 0, b = 0;
            if (row < rows) {
                a = wave_uniform(rowptr[row]);
                b = wave_uniform(rowptr[row + 1]);
            }
            acc[r][0] = row_direct(colidx, val, Bt, ld32, lane_off, lane, a, b);
        }
    } else {
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double x = acc[r][j];
                x += __shfl_xor(x, 16, WAVE);
                x += __shfl_xor(x, 32, WAVE);
                acc[r][j] = x;
            }
    }
    if (tid == 0 && blockIdx.y == 0) atomicAdd(&g_panel_stats[fell_back ? 2 : 0], 1ull);

    // park the panel as [column][row] in the (now dead) tile buffers and write it back along rows
    double *ctile = smem;
    if (fell_back) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) ctile[lane * (R + 1) + wave * RPW + r] = acc[r][0];
    } else if (lane < 16) {
        const int jj = lane;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int rr = wave * RPW + r;
            ctile[(2 * jj) * (R + 1) + rr] = acc[r][0];
            ctile[(2 * jj + 1) * (R + 1) + rr] = acc[r][1];
            ctile[(32 + 2 * jj) * (R + 1) + rr] = acc[r][2];
            ctile[(33 + 2 * jj) * (R + 1) + rr] = acc[r][3];
        }
    }
    __syncthreads(); // F
    const int nrows = min(R, rows - row0);
    const int ncols = min(64, n - col0);
    for (int idx = tid; idx < 64 * R; idx += 1024) {
        const int r = idx % R, j = idx / R;
        if (r < nrows && j < ncols) {
            double *dst = C + (int64_t)(col0 + j) * ldc + (row0 + r);
            const double sres = alpha * ctile[j * (R + 1) + r];
            *dst = (beta == 0.0) ? sres : fma(beta, *dst, sres);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 2, windowed form, sixth generation: one DPP row per MATRIX row.
//
// tools/visit_bench.hip (the visit skeleton of generations 4/5 on L2-resident data) shows two saturated units: the
// LDS read port (8 KB per sixteen slots = 32 cycles at 256 B/clk -- irreducible, 512 bytes of B per nonzero) and the
// vector-memory address unit (~15 cycles per wave instruction whatever it fetches: two window loads per visit plus
// one 1 KB tile-DMA instruction per visit).  This generation cuts the second: a "super-visit" handles FOUR matrix rows
// at once -- DPP row q of the wave is matrix row 4g+q, lane k of that row holds entries cursor_q + k (set A) and
// cursor_q + 16 + k (set B) of that row's window, step k broadcasts entry k of all four rows -- so
//   * four buffer loads fetch 32-entry windows of four rows (8 rows' worth of the old loads), issued right after
//     the super-visit into the registers it just finished with (no ring), a whole tile ahead of their use;
//   * a lane accumulates 4 columns of ONE matrix row: 8 accumulator registers per four rows instead of 32, which
//     pays for G = 2 groups per wave -> 128-row panels, half the tile DMA per nonzero (and no cross-row fold at the end);
//   * cursors, row ends, counts and the order check live in vector registers (one value per DPP row); the scalar
//     unit only sees the largest count of the four rows, which sets the number of 4-step blocks to run.
// The instruction mix of a step is unchanged (v_add_u32_dpp, two ds_read_b128, four v_fmac_f64_dpp).
// ---------------------------------------------------------------------------------------------
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
                 "v_cmp_gt_u32 vcc, 0x80, %[co]\n\t"
                 "v_lshlrev_b32 %[co], 9, %[co]\n\t"
                 "s_and_b64 vcc, vcc, %[m]\n\t"
                 "s_mov_b64 %[m], vcc\n\t"
                 "v_cndmask_b32 %[co], %[zr], %[co], vcc\n\t"
                 "v_cndmask_b32 %[glo], 0, %[vlo], vcc\n\t"
                 "v_cndmask_b32 %[ghi], 0, %[vhi], vcc"
                 : [co] "=&v"(co), [m] "=&s"(m), [glo] "=&v"(glo), [ghi] "=&v"(ghi)
                 : [tlo] "s"(tile_lo), [wc] "v"(wc), [rem] "v"(rem), [k] "v"(k), [zr] "v"(zero_rel),
                   [vlo] "v"(__double2loint(wv)), [vhi] "v"(__double2hiint(wv))
                 : "vcc", "scc");
    gv = __hiloint2double(ghi, glo);
}
// the four 32-entry windows of a group: col_idx and val, entries 0-15 (A) and 16-31 (B) of every row
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

// ABL: diagnostic switches compiled in (SBLAS_ABLATE: 0x10000 no LDS reads / FMAs, 0x20000 no tile DMA, 0x40000 no
// per-tile barrier, 0x40000000 no tile loop at all -- the fixed cost per panel: 48 us of 300 on the bench matrix;
// wrong results)
template <int G, bool ABL>
__global__ __launch_bounds__(1024) void spmm_window6_kernel(
    int rows, int cols, int npanels, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, const int2 *__restrict__ info, int panel_rows, int nnz, int ablate)
{
    const bool no_math = ABL && (ablate & 0x10000) != 0, no_dma = ABL && (ablate & 0x20000) != 0;
    const bool no_bar = ABL && (ablate & 0x40000) != 0;
    constexpr int RW = 4 * G, RMAX = 16 * RW;
    static_assert(G == 2 || G == 3, "two or three groups per wave (the counted vmcnt waits are 4 G)");
    static_assert(64 * (RMAX + 1) <= 2 * W2_TILE, "C tile must fit in the (dead) B tile buffers");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *zero_row = smem + 2 * W2_TILE;
    int *sm_i = reinterpret_cast<int *>(smem + 2 * W2_TILE + 64); // [0] = bad

    const int panel = xcd_contiguous_panel(blockIdx.x, npanels);
    const int2 span = info[panel];
    if (span.x > span.y) return; // the direct kernel owns this panel

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    const int row0 = panel * panel_rows;
    const int col0 = blockIdx.y * 64;
    const unsigned ld32 = (unsigned)ldbt;
    const int t_lo = span.x / W2_ROWS, t_hi = (ABL && (ablate & 0x40000000)) ? t_lo - 1 : span.y / W2_ROWS;
    // waves 12-15 load the tiles, waves 0-11 consume (a wave that did both had its window loads retire behind its own
    // tile fetches -- vmcnt is in order -- and the kernel ran twice as long)
    const bool loader = wave >= 12;
    const bool active = !loader && wave * RW < panel_rows; // this wave has rows

    if (tid < 64) zero_row[tid] = 0.0;
    if (tid == 0) sm_i[0] = 0;

    double acc[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[g][j] = 0.0;

    // ---- tile DMA: loader wave lw (= wave - 12) moves Bt rows 32 lw .. 32 lw + 31 of a tile: 16 instructions of two
    // rows each (lanes 0-31 the first, lanes 32-63 the second); tile row j lives at LDS byte j * 512.  Rows past the end
    // of B are clamped to row `cols`, the all-zero row of the workspace (scalar addresses when all 32 rows exist)
    const unsigned ldb8 = ld32 * 8u;
    const unsigned piece_off = (unsigned)(col0 + ((lane & 31) << 1)) * 8u; // bytes inside a Bt row
    const unsigned pair_off = (unsigned)(lane >> 5) * ldb8 + piece_off;
    const char *bt_bytes = reinterpret_cast<const char *>(Bt);
    int dummy = 0;
    // instructions i0 .. i1-1 of the wave's share
    auto dma_part = [&](int t, int buf, int i0, int i1) {
        const int lw = wave - 12;
        const int r0 = t * W2_ROWS + lw * 32;
        const unsigned lds0 = (unsigned)(uintptr_t)(smem + buf * W2_TILE) + (unsigned)lw * 16384u;
        if (no_dma) {
            // (read-write operand: the landing register must stay reserved while the loads are in flight)
            for (int i = i0; i < i1; ++i) asm volatile("global_load_dword %0, %1, off" : "+v"(dummy) : "v"(bt_bytes) : "memory");
        } else if (r0 + 31 <= cols) {
            const char *p = bt_bytes + (size_t)((unsigned)r0 * ldb8);
            for (int i = i0; i < i1; ++i) dma_rows_scalar(lds0 + i * 1024u, pair_off, p + (size_t)(2u * i) * ldb8);
        } else {
            for (int i = i0; i < i1; ++i) {
                const unsigned brow = (unsigned)min(r0 + 2 * i + (lane >> 5), cols);
                dma_rows_vector(lds0 + i * 1024u, bt_bytes + (size_t)(brow * ldb8 + piece_off));
            }
        }
    };

    // the first tile does not depend on the row pointers: fetch it while they are on their way
    if (loader) dma_part(t_lo, 0, 0, 16);

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
#pragma unroll
    for (int g = 0; g < G; ++g) window_issue6(rc, rv, cur[g] + k, wca[g], wva[g], wcb[g], wvb[g]);
    // everything lands before the loop starts, so its counted waits (written for the steady state) hold from the
    // first tile on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads(); // P
    for (int t = t_lo; t <= t_hi; ++t) {
        const int cbuf = (t - t_lo) & 1;
        // next tile (that buffer was last read before the previous barrier)
        if (loader && t < t_hi) dma_part(t + 1, cbuf ^ 1, 0, 16);
        const int tile_lo = t * W2_ROWS;
        const unsigned tile_base = (unsigned)(uintptr_t)(smem + cbuf * W2_TILE);
        const unsigned lb = tile_base + (unsigned)k * 16u;
        const unsigned zero_rel = (unsigned)(uintptr_t)zero_row - tile_base;
        auto visit = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            // younger than this group's windows: the other groups' windows (4 each)
            window_wait6<4 * (G - 1)>(wca[g], wva[g], wcb[g], wvb[g]);
            double &q0 = acc[g][0], &q1 = acc[g][1], &q2 = acc[g][2], &q3 = acc[g][3];
            for (;;) {
                const int rem = end[g] - cur[g];
                unsigned coA, coB;
                double gvA, gvB;
                unsigned long long mA, mB;
                window_select6(wca[g], wva[g], tile_lo, rem, k, zero_rel, coA, gvA, mA);
                window_select6(wcb[g], wvb[g], tile_lo, rem - 16, k, zero_rel, coB, gvB, mB);
                // entries of this tile per matrix row (= per DPP row), as a per-lane value
                const unsigned fa = q < 2 ? (unsigned)mA : (unsigned)(mA >> 32);
                const unsigned fb = q < 2 ? (unsigned)mB : (unsigned)(mB >> 32);
                const int sh = (q & 1) * 16;
                const int take = __popc((fa >> sh) & 0xffffu) + __popc((fb >> sh) & 0xffffu);
                // with ascending columns they are exactly the first `take` entries of the row's window
                viol |= mA ^ __builtin_amdgcn_ballot_w64(k < take);
                viol |= mB ^ __builtin_amdgcn_ballot_w64(k + 16 < take);
                int mx = max(max(__builtin_amdgcn_readlane(take, 0), __builtin_amdgcn_readlane(take, 16)),
                             max(__builtin_amdgcn_readlane(take, 32), __builtin_amdgcn_readlane(take, 48)));
                if (no_math) mx = 0;
                // (blocks of eight steps with the second half's LDS reads issued ahead of the first half's FMAs were
                //  tried: no gain)
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
                cur[g] += take;
                // a row that used its whole window and has more: fetch the next windows now and go again (rare:
                // more than 32 nonzeros of a row inside one 128-column tile)
                const bool more = take >= 32 && cur[g] < end[g];
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(more) == 0ull, 1)) break;
                window_issue6(rc, rv, cur[g] + k, wca[g], wva[g], wcb[g], wvb[g]);
                window_wait6<0>(wca[g], wva[g], wcb[g], wvb[g]); // drains the queue: later counted waits stay correct
            }
            // next tile's windows, into the registers this super-visit is done with
            window_issue6(rc, rv, cur[g] + k, wca[g], wva[g], wcb[g], wvb[g]);
        };
        if (active) {
            visit(std::integral_constant<int, 0>{});
            visit(std::integral_constant<int, 1>{});
            if constexpr (G > 2) visit(std::integral_constant<int, 2>{});
        }
        if (loader) asm volatile("s_waitcnt vmcnt(0)" : "+v"(dummy)::"memory"); // the tile has landed
        if (!no_bar) __syncthreads(); // E_t
    }
#pragma unroll
    for (int g = 0; g < G; ++g) window_wait6<0>(wca[g], wva[g], wcb[g], wvb[g]); // retire the unused last fetches
    int bad = viol != 0ull ? 1 : 0;
#pragma unroll
    for (int g = 0; g < G; ++g)
        if (__builtin_amdgcn_ballot_w64(cur[g] < end[g]) != 0ull) bad = 1; // unconsumed nonzeros
    if (ABL && ablate != 0) bad = 0;
    if (bad && lane == 0) atomicOr(&sm_i[0], 1);
    __syncthreads(); // V
    const bool fell_back = sm_i[0] != 0;
    if (fell_back) {
        // recompute straight from L2, one column per lane: acc[g][j] <- row 4g+j of the wave
        const unsigned lane_off = (unsigned)(col0 + lane);
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
                acc[g][j] = row_direct(colidx, val, Bt, ld32, lane_off, lane, a, b);
            }
    }
    if (tid == 0 && blockIdx.y == 0) atomicAdd(&g_panel_stats[fell_back ? 2 : 0], 1ull);

    // park the panel as [column][row] in the (now dead) tile buffers and write it back along rows
    double *ctile = smem;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (fell_back) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ctile[lane * (RMAX + 1) + wave * RW + 4 * g + j] = acc[g][j];
        } else {
            const int rr = wave * RW + 4 * g + q;
            ctile[(2 * k) * (RMAX + 1) + rr] = acc[g][0];
            ctile[(2 * k + 1) * (RMAX + 1) + rr] = acc[g][1];
            ctile[(32 + 2 * k) * (RMAX + 1) + rr] = acc[g][2];
            ctile[(33 + 2 * k) * (RMAX + 1) + rr] = acc[g][3];
        }
    }
    __syncthreads(); // F
    const int nrows = min(panel_rows, rows - row0);
    const int ncols = min(64, n - col0);
    // (fetching the old C values in the prologue, to take their HBM latency out of the epilogue, did not pay: the
    //  registers they hold across the tile loop spill)
    for (int idx = tid; idx < 64 * panel_rows; idx += 1024) {
        const int r = idx % panel_rows, j = idx / panel_rows;
        if (r < nrows && j < ncols) {
            double *dst = C + (int64_t)(col0 + j) * ldc + (row0 + r);
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
    double *__restrict__ C, int64_t ldc, const int2 *__restrict__ info, int info_panel_rows, int interleave,
    int epoch)
{
    static_assert(GROUPS == 1 || GROUPS == 2 || GROUPS == 4, "lane groups of 64, 32 or 16 lanes");
    constexpr int TILE_COLS = 128 / GROUPS;
    constexpr int PER_STEP = 16 * GROUPS; // nonzeros handled by one 16-slot sweep
    constexpr int GLANES = 64 / GROUPS;   // lanes that share a nonzero
    __shared__ double ctile[TILE_COLS][WIDE_PANEL + 1];
    // every panel windowed (the bench matrix): one scalar load of one shared address and out, instead of two
    // dependent loads per workgroup (4500 workgroups of early exits took 16 us of a 340 us step)
    if (info != nullptr && info[(rows + info_panel_rows - 1) / info_panel_rows + 1].x != epoch) return;
    const int lane = threadIdx.x & 63;
    const int wave = wave_uniform(threadIdx.x >> 6);
    // interleave: neighbouring panels on different XCDs, so that the whole chip sweeps one band of B at a time (wide
    // bands: the band must fit the Infinity Cache once, not once per XCD); otherwise one contiguous range per XCD
    // (`interleave` < 0: decide from the column span the classifier recorded -- a band of B rows wider than 16 MB)
    if (interleave < 0) {
        interleave = 0;
        if (info != nullptr) { // one value for every workgroup: the middle panel's span, left by the classifier
            const int band = info[(rows + info_panel_rows - 1) / info_panel_rows].x;
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
    if (info && mine) {
        const int2 span = info[row / info_panel_rows];
        mine = span.x > span.y;
        if (mine && lane == 0 && blockIdx.y == 0 && row % info_panel_rows == 0) atomicAdd(&g_panel_stats[1], 1ull);
    }
    const int sub = lane & 15;
    const int half = lane / GLANES; // which of the GROUPS nonzeros of a step this lane works on
    const unsigned ldb8 = (unsigned)ldbt * 8u;                                       // bytes per Bt row
    const unsigned lb = (unsigned)(col0 * 8) + (unsigned)(lane % GLANES) * 16u; // this lane's 2 columns
    const unsigned zero_off = (unsigned)cols * ldb8;                                 // Bt[cols][*] == 0
    const char *__restrict__ bt_bytes = reinterpret_cast<const char *>(Bt);

    double acc0 = 0.0, acc1 = 0.0;
    if (mine) {
        const int p0 = wave_uniform(rowptr[row]);
        const int p1 = wave_uniform(rowptr[row + 1]);
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
    }
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
                                                                  double *__restrict__ C, int64_t ldc)
{
    constexpr int GROUPS = 256 / G;
    constexpr int RPG = PANEL_ROWS / GROUPS; // rows per group
    static_assert(RPG >= 1, "panel too small for this group width");
    __shared__ double ctile[G][PANEL_ROWS + 1];
    const int l = threadIdx.x % G, grp = threadIdx.x / G;
    const int row0 = blockIdx.x * PANEL_ROWS;
    for (int rr = 0; rr < RPG; ++rr) {
        const int r = grp * RPG + rr;
        const int row = row0 + r;
        double acc = 0.0;
        if (row < rows) {
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
    }
    __syncthreads();
    const int nrows = min(PANEL_ROWS, rows - row0);
    for (int idx = threadIdx.x; idx < G * PANEL_ROWS; idx += 256) {
        const int r = idx % PANEL_ROWS, j = idx / PANEL_ROWS;
        if (r < nrows && j < n) {
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
// Used for the direct panels when the matrix averages fewer than 32 nonzeros per row.
// ---------------------------------------------------------------------------------------------
constexpr int ROWS_PANEL = 64; // rows per workgroup: 16 waves x 4
__global__ __launch_bounds__(1024) void spmm_direct_rows_kernel(
    int rows, int cols, int npanels, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, const int2 *__restrict__ info, int info_panel_rows, int interleave,
    int epoch)
{
    __shared__ double ctile[64][ROWS_PANEL + 1];
    __shared__ int row_mine[ROWS_PANEL];
    if (info != nullptr && info[(rows + info_panel_rows - 1) / info_panel_rows + 1].x != epoch) return; // nothing direct
    const int lane = threadIdx.x & 63;
    const int wave = wave_uniform(threadIdx.x >> 6);
    const int k = lane & 15, q = lane >> 4;
    const int col0 = blockIdx.y * 64;
    if (col0 >= n) return; // a padding tile of ldbt (uniform over the workgroup)
    if (interleave < 0) {
        interleave = 0;
        if (info != nullptr) {
            const int band = info[(rows + info_panel_rows - 1) / info_panel_rows].x;
            interleave = (long long)band * 512 > (16ll << 20);
        }
    }
    const int row0 = (interleave ? (int)blockIdx.x : xcd_contiguous_panel(blockIdx.x, npanels)) * ROWS_PANEL;
    const int rr = wave * 4 + q;
    const int row = row0 + rr;
    bool mine = row < rows;
    if (info && mine) {
        const int2 span = info[row / info_panel_rows];
        mine = span.x > span.y;
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
    const int nrows = min(ROWS_PANEL, rows - row0);
    const int ncols = min(64, n - col0);
    for (int idx = threadIdx.x; idx < 64 * ROWS_PANEL; idx += 1024) {
        const int r = idx % ROWS_PANEL, j = idx / ROWS_PANEL;
        if (r < nrows && j < ncols && row_mine[r]) {
            double *dst = C + (int64_t)(col0 + j) * ldc + (row0 + r);
            const double sres = alpha * ctile[j][r];
            *dst = (beta == 0.0) ? sres : fma(beta, *dst, sres);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SpMV: LPR lanes per row (a power of two, 4..64), 256/LPR rows per workgroup.  The lanes of a
// group stride through the row's nonzeros (coalesced col_idx / val streams, x gathered through
// L2), then the partial sums are folded with xor-shuffles inside the wave -- the wave64 successor of
// the reference's unused sum_32_shfl (utility.h:241-246).
// ---------------------------------------------------------------------------------------------
template <int LPR, bool GATHER = true> // GATHER = false: diagnostic build that reads x[lane] (wrong results)
__global__ __launch_bounds__(256) void spmv_csr_kernel(int rows, const int *__restrict__ rowptr,
                                                      const int *__restrict__ colidx,
                                                      const double *__restrict__ val,
                                                      const double *__restrict__ x, double alpha, double beta,
                                                      double *__restrict__ y)
{
    constexpr int ROWS_PER_BLOCK = 256 / LPR;
    const int l = threadIdx.x % LPR;
    const int row = blockIdx.x * ROWS_PER_BLOCK + threadIdx.x / LPR;
    double s0 = 0.0, s1 = 0.0;
    if (row < rows) {
        const int p1 = rowptr[row + 1];
        int p = rowptr[row] + l;
        // two slices per trip (four gave fewer resident waves and ran 20 % slower)
        for (; p + LPR < p1; p += 2 * LPR) {
            const int c0 = colidx[p], c1 = colidx[p + LPR];
            const double a0 = val[p], a1 = val[p + LPR];
            s0 = fma(a0, x[GATHER ? c0 : (c0 & 63)], s0);
            s1 = fma(a1, x[GATHER ? c1 : (c1 & 63)], s1);
        }
        if (p < p1) s0 = fma(val[p], x[colidx[p]], s0);
    }
    double s = s0 + s1;
#pragma unroll
    for (int m = LPR / 2; m > 0; m >>= 1) s += __shfl_xor(s, m, WAVE);
    if (row < rows && l == 0) {
        const double r = alpha * s;
        y[row] = (beta == 0.0) ? r : fma(beta, y[row], r);
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
                                                        double beta, double *__restrict__ C, int64_t ldc)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave_uniform(threadIdx.x >> 6);
    if (row >= rows) return;
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
// SpMV for short and medium rows (5..48 nonzeros per row on average), stream form.  The lanes-per-row kernel gives every row a lane group
// of 4..32 lanes: a 5-nonzero row keeps 4 of 8 lanes busy for two trips, and the stencil-like matrices that have such
// rows run at 2.6-3.3 TB/s.  Here a 256-thread block owns 256 consecutive rows, i.e. ONE contiguous run of nonzeros:
// all threads stream it (thread t takes nonzeros t, t + 256, ...; every lane busy, fully coalesced), park the
// products in LDS, and thread r then adds up the products of row r in CSR order.  A block whose rows hold more than
// the LDS can take (longer rows among the short ones) takes its rows in several runs.
// ---------------------------------------------------------------------------------------------
constexpr int ST_ROWS = 256;
constexpr int ST_CAP = 6144; // products per block (48 KiB + skew): three blocks per CU
__device__ __forceinline__ int st_skew(int q) { return q + (q >> 5); } // rows of equal length: spread the LDS banks
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
        if (tid >= r0 && tid < r1) {
            double sum = 0.0;
            for (int q = sp[tid] - base, e = sp[tid + 1] - base; q < e; ++q) sum += prod[st_skew(q)];
            const double res = alpha * sum;
            y[row0 + tid] = (beta == 0.0) ? res : fma(beta, y[row0 + tid], res);
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
// instruction, more for a 40-line gather).  Here a 16-row block (one row per wave) fetches the x range its rows span
// into LDS once and gathers from there; the stream loads of a row (up to 448 nonzeros) are issued right after its row
// pointers, BEFORE the window is known, so the block-wide min/max, the window load and their three barriers hide
// behind the HBM latency of the stream (the barriers are `s_barrier` without the vmcnt(0) of __syncthreads).
// Columns outside the window (unsorted rows) are fetched from global memory lane by lane.
// ---------------------------------------------------------------------------------------------
constexpr int SPMV_LDS_ROWS = 16;   // = waves per block
constexpr int SPMV_LDS_CAP = 5120;  // doubles (40 KiB): two blocks per CU
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int RW, int S> // RW rows per wave (16 RW rows per block), S slices of 64 nonzeros fetched ahead per row
__global__ __launch_bounds__(1024) void spmv_csr_lds_kernel(int rows, int cols, const int *__restrict__ rowptr,
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
        for (int p0 = wave * 64; p0 < pairs; p0 += 1024) { // (wave-uniform trip count)
            const int pr = min(p0 + lane, pairs - 1);  // clamped lanes rewrite the last pair into the slack area
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)pr * 16),
                                             (__attribute__((address_space(3))) void *)(xs + 2 * p0), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (wlen > 0) {
        // the whole window in one burst of loads (clamped indices), then the stores
        constexpr int PASSES = SPMV_LDS_CAP / 1024;
        double t[PASSES];
#pragma unroll
        for (int j = 0; j < PASSES; ++j) t[j] = x[lo + min(tid + 1024 * j, wlen - 1)];
#pragma unroll
        for (int j = 0; j < PASSES; ++j)
            if (tid + 1024 * j < wlen) xs[tid + 1024 * j] = t[j];
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
                    xv[u] = xs[inw ? rel : 0u];
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

// ---------------------------------------------------------------------------------------------
// SpMV for long rows, flat form.  The lanes-per-row kernel above walks a 400-nonzero row in four dependent trips of
// {col/val load -> x gather -> FMA} with at most 1.5 KB in flight per wave.  Here a wave issues the (col, val) loads
// of S slices of 64 nonzeros back to back, then all gathers, then the FMAs -- two memory round trips per S*64
// nonzeros -- with NO predication: indices are clamped to the last nonzero of the row and the values of the clamped
// lanes are zeroed afterwards (the round-1 "burst" kernel predicated every load and hipcc put a full vmcnt(0)
// between them: slower than the plain kernel).
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void spmv_csr_flat_kernel(int rows, const int *__restrict__ rowptr,
                                                           const int *__restrict__ colidx,
                                                           const double *__restrict__ val,
                                                           const double *__restrict__ x, double alpha, double beta,
                                                           double *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int p0 = wave_uniform(rowptr[row]), p1 = wave_uniform(rowptr[row + 1]);
    double s0 = 0.0, s1 = 0.0;
    const int last = p1 - 1;
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
            const double av = (base + u * WAVE + lane <= last) ? a[u] : 0.0;
            if (u & 1) s1 = fma(av, xv[u], s1);
            else s0 = fma(av, xv[u], s0);
        }
    }
    double sum = s0 + s1;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m, WAVE);
    if (lane == 0) {
        const double r = alpha * sum;
        y[row] = (beta == 0.0) ? r : fma(beta, y[row], r);
    }
}

// ---------------------------------------------------------------------------------------------
// SpMV for long rows, burst form.  The generic kernel is latency-bound (SQ_WAIT_ANY 86 %, ~650 cycles per
// L1->L2 request): a 400-nonzero row walks four dependent trips of {col/val load -> x gather -> FMA}.  Here a wave
// issues the (col, val) loads of up to 512 nonzeros of its row back to back, then all x gathers, then the FMAs:
// one memory round trip per stage per 512 nonzeros, 16 independent loads in flight per lane.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spmv_csr_burst_kernel(int rows, const int *__restrict__ rowptr,
                                                            const int *__restrict__ colidx,
                                                            const double *__restrict__ val,
                                                            const double *__restrict__ x, double alpha, double beta,
                                                            double *__restrict__ y)
{
    constexpr int S = 8; // slices of 64 nonzeros per burst
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int p0 = wave_uniform(rowptr[row]), p1 = wave_uniform(rowptr[row + 1]);
    double s0 = 0.0, s1 = 0.0;
    for (int base = p0; base < p1; base += S * WAVE) {
        int c[S];
        double a[S], xv[S];
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int p = base + u * WAVE + lane;
            c[u] = 0;
            a[u] = 0.0;
            if (p < p1) {
                c[u] = colidx[p];
                a[u] = val[p];
            }
        }
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int p = base + u * WAVE + lane;
            xv[u] = (p < p1) ? x[c[u]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < S; u += 2) {
            s0 = fma(a[u], xv[u], s0);
            s1 = fma(a[u + 1], xv[u + 1], s1);
        }
    }
    double sum = s0 + s1;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m, WAVE);
    if (lane == 0) {
        const double r = alpha * sum;
        y[row] = (beta == 0.0) ? r : fma(beta, y[row], r);
    }
}

// ---------------------------------------------------------------------------------------------
// SpMV with the x window of a row block staged in LDS.
// The plain kernel is bound by the address unit, not by HBM: per 64 nonzeros it issues two coalesced loads
// (col_idx, val) and one 64-address gather of x, and the gather keeps the unit busy about twice as long as both
// streams together.  Here a workgroup of 8 waves owns 16 consecutive rows (two per wave, so the CU still holds 32
// waves), guesses their column window from the first/last column of each row, copies x[lo, hi] into LDS with
// coalesced loads and gathers from LDS.  Any column outside the staged window -- unsorted rows, outliers, windows
// larger than the LDS budget -- is read from global memory by that lane: the result never depends on the guess.
// ---------------------------------------------------------------------------------------------
constexpr int SPMV_BLOCK_ROWS = 16;
constexpr int SPMV_WINDOW_CAP = 4608; // doubles (36 KiB): four workgroups = 32 waves per CU

__global__ __launch_bounds__(512) void spmv_csr_window_kernel(int rows, int cols, const int *__restrict__ rowptr,
                                                             const int *__restrict__ colidx,
                                                             const double *__restrict__ val,
                                                             const double *__restrict__ x, double alpha, double beta,
                                                             double *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) double xs[];
    __shared__ int sm_lo, sm_hi;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    const int row0 = blockIdx.x * SPMV_BLOCK_ROWS;
    if (tid == 0) {
        sm_lo = 0x7fffffff;
        sm_hi = -1;
    }
    // this wave's two rows (their pointers are fetched before the window is known)
    const int rowA = row0 + wave * 2, rowB = rowA + 1;
    int pa0 = 0, pa1 = 0, pb1 = 0;
    if (rowA < rows) {
        pa0 = wave_uniform(rowptr[rowA]);
        pa1 = wave_uniform(rowptr[rowA + 1]);
        pb1 = (rowB < rows) ? wave_uniform(rowptr[rowB + 1]) : pa1;
    }
    __syncthreads();
    if (lane == 0 && pb1 > pa0) {
        int first = (pa1 > pa0) ? colidx[pa0] : colidx[pa1];
        int last = (pb1 > pa1) ? colidx[pb1 - 1] : colidx[pa1 - 1];
        if (pa1 > pa0 && pb1 > pa1) { // both rows non-empty: the window must cover both ends of both
            first = min(first, colidx[pa1]);
            last = max(last, colidx[pa1 - 1]);
        }
        atomicMin(&sm_lo, first);
        atomicMax(&sm_hi, last);
    }
    __syncthreads();
    int lo = sm_lo, hi = sm_hi;
    if (lo > hi) {
        lo = 0;
        hi = -1;
    }
    if (lo < 0) lo = 0;
    if (hi >= cols) hi = cols - 1;
    int wlen = hi - lo + 1;
    if (wlen > SPMV_WINDOW_CAP) wlen = SPMV_WINDOW_CAP; // keep the first part of an oversized window
    for (int i = tid; i < wlen; i += 512) xs[i] = x[lo + i];
    __syncthreads();

#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const int row = which ? rowB : rowA;
        const int p0 = which ? pa1 : pa0, p1 = which ? pb1 : pa1;
        if (row >= rows) break;
        double s0 = 0.0, s1 = 0.0;
        int p = p0 + lane;
        for (; p + WAVE < p1; p += 2 * WAVE) {
            const int c0 = colidx[p], c1 = colidx[p + WAVE];
            const double a0 = val[p], a1 = val[p + WAVE];
            const unsigned r0 = (unsigned)(c0 - lo), r1 = (unsigned)(c1 - lo);
            const double x0 = (r0 < (unsigned)wlen) ? xs[r0] : x[c0];
            const double x1 = (r1 < (unsigned)wlen) ? xs[r1] : x[c1];
            s0 = fma(a0, x0, s0);
            s1 = fma(a1, x1, s1);
        }
        if (p < p1) {
            const int c0 = colidx[p];
            const unsigned r0 = (unsigned)(c0 - lo);
            const double x0 = (r0 < (unsigned)wlen) ? xs[r0] : x[c0];
            s0 = fma(val[p], x0, s0);
        }
        double sum = s0 + s1;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m, WAVE);
        if (lane == 0) {
            const double r = alpha * sum;
            y[row] = (beta == 0.0) ? r : fma(beta, y[row], r);
        }
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

// In-place sum of g replicas that share a device: every buffer ends up holding the sum, added in
// rank order (the single-device stand-in for the all-reduce when ranks are oversubscribed).
__global__ __launch_bounds__(256) void sum_replicas_kernel(ReplicaPtrs bufs, int g, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double s = 0.0;
        for (int q = 0; q < g; ++q) s += bufs.p[q][i];
        for (int q = 0; q < g; ++q) bufs.p[q][i] = s;
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

hipError_t launch_dense_to_rowmajor(hipStream_t s, int64_t cols, int64_t n, const double *B, int64_t ldb,
                                    double *Bt, int64_t ldbt)
{
    dim3 grid((unsigned)((cols + 1 + STAGE_K - 1) / STAGE_K), (unsigned)((ldbt + 63) / 64));
    hipLaunchKernelGGL(dense_to_rowmajor_kernel, grid, dim3(256), 0, s, cols, n, B, ldb, Bt, ldbt);
    return hipGetLastError();
}

// compute units of the current device (queried once per device)
static int compute_units()
{
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cached[dev] <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// Diagnostics: two events around the dominant stage-2 kernel of the most recent launch on a device (see
// sblas_hip_debug_spmm_kernel_events).
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

// Sixth generation, one workgroup per CU at a time: the groups per wave (2 or 3) and the panel height (a multiple of
// the rows of a wave) that minimise rounds x (height + per-tile fixed cost).
static void gen6_plan(int rows, int &info_rows, int &gen6_g)
{
    const int ncu = compute_units();
    int best = 128;
    long best_cost = -1;
    gen6_g = 2;
    for (int g = 2; g <= W6_GMAX; ++g)
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
    const char *pr = getenv("SBLAS_SPMM_PANEL_ROWS"); /* experiments: "<rows>" or "<rows>,<groups>" */
    if (pr) {
        int r = atoi(pr), g = strchr(pr, ',') ? atoi(strchr(pr, ',') + 1) : (r % 12 == 0 && r > 128 ? 3 : 2);
        if ((g == 2 || g == 3) && r >= SPMM_MIN_PANEL_ROWS && r <= 48 * g && r % (4 * g) == 0) { // (workspace: a verdict per 32 rows)
            info_rows = r;
            gen6_g = g;
        }
    }
}

static std::atomic<int> g_epoch{1}; // tags one call's classifier verdicts (see classify_panel)

// Stage 1 + classifier of the default (sixth-generation) path in one launch; the epoch goes to launch_spmm_rowpanel.
hipError_t launch_stage_classify(hipStream_t s, int64_t cols, int64_t n, const double *B, int64_t ldb, double *Bt,
                                 int64_t ldbt, int rows, const int *rowptr, const int *colidx, int *epoch_out)
{
    int info_rows = 0, g = 2;
    gen6_plan(rows, info_rows, g);
    const int np = (rows + info_rows - 1) / info_rows;
    const int stage_blocks = (int)((cols + 1 + STAGE_K - 1) / STAGE_K);
    const int epoch = g_epoch.fetch_add(1, std::memory_order_relaxed);
    int2 *winfo = reinterpret_cast<int2 *>(Bt + (size_t)(cols + 1) * (size_t)ldbt);
    dim3 grid((unsigned)(stage_blocks + (np + 3) / 4), (unsigned)((ldbt + 63) / 64));
    hipLaunchKernelGGL(stage_classify_kernel, grid, dim3(256), 0, s, cols, n, B, ldb, Bt, ldbt, stage_blocks, rows, np,
                       info_rows, rowptr, colidx, 1 << 24, (float)info_rows / 16.0f, winfo, epoch);
    *epoch_out = epoch;
    return hipGetLastError();
}

hipError_t launch_spmm_rowpanel(hipStream_t s, int rows, int cols, int64_t nnz, const int *rowptr, const int *colidx,
                                const double *val, const double *Bt, int64_t ldbt, int n, double alpha,
                                double beta, double *C, int64_t ldc, int variant, int pre_epoch)
{
    const unsigned panels = (unsigned)((rows + PANEL_ROWS - 1) / PANEL_ROWS);
    if (ldbt >= 64) {
        if (variant == SPMM_VARIANT_DIRECT_DPP || variant == SPMM_VARIANT_DIRECT_ROWS || variant == SPMM_VARIANT_AUTO ||
            variant == SPMM_VARIANT_WINDOW2 ||
            variant == SPMM_VARIANT_WINDOW3 || variant == SPMM_VARIANT_WINDOW4 || variant == SPMM_VARIANT_WINDOW5 ||
            variant == SPMM_VARIANT_WINDOW6) {
            const int2 *info = nullptr;
            int info_rows = 1;
            // pre_epoch != 0: launch_stage_classify has classified the panels already (default variant only)
            const bool preclassified = pre_epoch != 0 && variant == SPMM_VARIANT_AUTO;
            const int epoch = preclassified ? pre_epoch : g_epoch.fetch_add(1, std::memory_order_relaxed);
            if (variant != SPMM_VARIANT_DIRECT_DPP && variant != SPMM_VARIANT_DIRECT_ROWS) {
                // 1. classify row panels; 2. windowed kernel on the qualifying ones; 3. direct kernel on the rest
                const bool gen2 = (variant == SPMM_VARIANT_WINDOW2);
                const bool gen4 = (variant == SPMM_VARIANT_WINDOW4);
                const bool gen5 = (variant == SPMM_VARIANT_WINDOW5);
                const bool gen6 = (variant == SPMM_VARIANT_WINDOW6 || variant == SPMM_VARIANT_AUTO);
                info_rows = gen2 ? W2_PANEL : gen4 ? W4_PANEL : gen5 ? W5_PANEL : W3_PANEL;
                int gen6_g = 2;
                if (gen6) gen6_plan(rows, info_rows, gen6_g);
                int2 *winfo = reinterpret_cast<int2 *>(const_cast<double *>(Bt) + (size_t)(cols + 1) * (size_t)ldbt);
                const int np = (rows + info_rows - 1) / info_rows;
                const double avg = rows > 0 ? (double)nnz / (double)rows : 0.0;
                const int need = (int)(avg * 1.15 / 64.0) + 1;
                const int ch = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 7;
                if (!preclassified)
                hipLaunchKernelGGL(classify_panels_kernel, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, s, rows, cols,
                                   np, info_rows, rowptr, colidx,
                                   /* generation 6 addresses a wave's 8-12 rows through 32-bit buffer offsets */
                                   gen6 ? (1 << 24) : (gen4 || gen5) ? 0x7fffffff : ch * 64,
                                   /* streaming generations: a (row, tile) visit costs what ~8 nonzeros cost in the
                                      direct kernel, so ask for 8 per row and 128-column tile on average */
                                   (gen4 || gen5 || gen6) ? (float)info_rows / 16.0f : 1.0f, winfo, gen5 ? 1 : 0, epoch);
                dim3 wgrid((unsigned)np, (unsigned)(ldbt / 64));
                const char *ab = getenv("SBLAS_ABLATE"); /* diagnostics only: wrong results when set */
                const int ablate = ab ? atoi(ab) : 0;
#define SBLAS_W_LAUNCH(KERNEL, CHV)                                                                                  \
    do {                                                                                                             \
        /* per device and cheap: set on every launch (one process may drive several GPUs) */                        \
        (void)hipFuncSetAttribute((const void *)KERNEL<CHV>, hipFuncAttributeMaxDynamicSharedMemorySize,             \
                                  (int)W2_LDS_BYTES);                                                                \
        hipLaunchKernelGGL(KERNEL<CHV>, wgrid, dim3(1024), W2_LDS_BYTES, s, rows, cols, np, rowptr, colidx, val, Bt, \
                           ldbt, n, alpha, beta, C, ldc, winfo, ablate);                                             \
    } while (0)
                KernelEvents *kev = kernel_events_slot();
                if (kev) (void)hipEventRecord(kev->a, s);
                if (gen6) {
#define SBLAS_W6_LAUNCH(GV, ABLV, ABLARG)                                                                             \
    do {                                                                                                             \
        const size_t lds_bytes = W2_LDS_BYTES + (ABLV ? 20480 : 0); /* diagnostics: scratch area for the DMA */        \
        (void)hipFuncSetAttribute((const void *)spmm_window6_kernel<GV, ABLV>,                                       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                       \
        hipLaunchKernelGGL((spmm_window6_kernel<GV, ABLV>), wgrid, dim3(1024), lds_bytes, s, rows, cols, np,         \
                           rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, winfo, info_rows, (int)nnz,        \
                           ABLARG);                                                                                  \
    } while (0)
                    if (ablate != 0) {
                        if (gen6_g == 3) SBLAS_W6_LAUNCH(3, true, ablate);
                        else SBLAS_W6_LAUNCH(2, true, ablate);
                    } else {
                        if (gen6_g == 3) SBLAS_W6_LAUNCH(3, false, 0);
                        else SBLAS_W6_LAUNCH(2, false, 0);
                    }
#undef SBLAS_W6_LAUNCH
                } else if (gen5) {
                    if (ablate != 0) {
                        (void)hipFuncSetAttribute((const void *)spmm_window5_kernel<true>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES);
                        hipLaunchKernelGGL(spmm_window5_kernel<true>, wgrid, dim3(1024), W2_LDS_BYTES, s, rows, cols,
                                           np, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, winfo, ablate,
                                           (int)nnz);
                    } else {
                        (void)hipFuncSetAttribute((const void *)spmm_window5_kernel<false>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES);
                        hipLaunchKernelGGL(spmm_window5_kernel<false>, wgrid, dim3(1024), W2_LDS_BYTES, s, rows, cols,
                                           np, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, winfo, 0,
                                           (int)nnz);
                    }
                } else if (gen4 && ablate != 0) {
                    (void)hipFuncSetAttribute((const void *)spmm_window4_kernel<true>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES);
                    hipLaunchKernelGGL(spmm_window4_kernel<true>, wgrid, dim3(1024), W2_LDS_BYTES, s, rows, cols, np,
                                       rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, winfo, ablate, (int)nnz);
                } else if (gen4) {
                    (void)hipFuncSetAttribute((const void *)spmm_window4_kernel<false>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES);
                    hipLaunchKernelGGL(spmm_window4_kernel<false>, wgrid, dim3(1024), W2_LDS_BYTES, s, rows, cols, np,
                                       rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, winfo, 0, (int)nnz);
                } else if (gen2) {
                    if (ch == 1) SBLAS_W_LAUNCH(spmm_window2_kernel, 1);
                    else if (ch == 2) SBLAS_W_LAUNCH(spmm_window2_kernel, 2);
                    else if (ch == 4) SBLAS_W_LAUNCH(spmm_window2_kernel, 4);
                    else SBLAS_W_LAUNCH(spmm_window2_kernel, 7);
                } else {
                    if (ch == 1) SBLAS_W_LAUNCH(spmm_window3_kernel, 1);
                    else if (ch == 2) SBLAS_W_LAUNCH(spmm_window3_kernel, 2);
                    else if (ch == 4) SBLAS_W_LAUNCH(spmm_window3_kernel, 4);
                    else SBLAS_W_LAUNCH(spmm_window3_kernel, 7);
                }
#undef SBLAS_W_LAUNCH
                if (kev) {
                    (void)hipEventRecord(kev->b, s);
                    kev->recorded = true;
                }
                info = winfo;
            }
            const int wide_panels = (rows + WIDE_PANEL - 1) / WIDE_PANEL;
            // experiments: unused dynamic LDS limits the resident workgroups per CU (rows in flight vs L2 reach)
            const char *dl = getenv("SBLAS_DIRECT_LDS");
            // 128-column tiles: one workgroup per CU (Queen-like rows at N = 256: +13 %, banded matrix at N = 128: +3 %)
            const size_t pad = dl ? (size_t)atoi(dl) : (ldbt == 64 ? 0 : 90000);
            const char *dm = getenv("SBLAS_DIRECT_MAP"); /* experiments: interleave | contiguous; default: by span */
            const int interleave = (dm && !strcmp(dm, "interleave")) ? 1 : (dm && !strcmp(dm, "contiguous")) ? 0 : -1;
            const double avg_row = rows > 0 ? (double)nnz / (double)rows : 0.0;
            if (n > 32 && (variant == SPMM_VARIANT_DIRECT_ROWS || (variant != SPMM_VARIANT_DIRECT_DPP && avg_row < 32.0))) {
                // short rows: four rows per wave
                const int rp = (rows + ROWS_PANEL - 1) / ROWS_PANEL;
                hipLaunchKernelGGL(spmm_direct_rows_kernel, dim3((unsigned)rp, (unsigned)(ldbt / 64)), dim3(1024), 0, s, rows,
                                   cols, rp, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, info, info_rows, interleave, epoch);
            } else if (ldbt == 64 && n <= 32) {
                dim3 grid((unsigned)wide_panels, 1u);
                if (pad) (void)hipFuncSetAttribute((const void *)spmm_direct_dpp_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
                hipLaunchKernelGGL(spmm_direct_dpp_kernel<4>, grid, dim3(WIDE_WAVES * 64), pad, s, rows, cols,
                                   wide_panels, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, info, info_rows, interleave, epoch);
            } else if (ldbt == 64) {
                dim3 grid((unsigned)wide_panels, 1u);
                if (pad) (void)hipFuncSetAttribute((const void *)spmm_direct_dpp_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
                hipLaunchKernelGGL(spmm_direct_dpp_kernel<2>, grid, dim3(WIDE_WAVES * 64), pad, s, rows, cols,
                                   wide_panels, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, info, info_rows, interleave, epoch);
            } else {
                dim3 grid((unsigned)wide_panels, (unsigned)(ldbt / 128));
                if (pad) (void)hipFuncSetAttribute((const void *)spmm_direct_dpp_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
                hipLaunchKernelGGL(spmm_direct_dpp_kernel<1>, grid, dim3(WIDE_WAVES * 64), pad, s, rows, cols,
                                   wide_panels, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, info, info_rows, interleave, epoch);
            }
        } else if (variant == SPMM_VARIANT_DIRECT) {
            const int wide_panels = (rows + WIDE_PANEL - 1) / WIDE_PANEL;
            dim3 grid((unsigned)wide_panels, (unsigned)(ldbt / 64));
            hipLaunchKernelGGL(spmm_rowpanel_kernel, grid, dim3(WIDE_WAVES * 64), 0, s, rows, wide_panels, rowptr,
                               colidx, val, Bt, ldbt, n, alpha, beta, C, ldc);
        } else {
            const float dens = 2.0f; // window a panel when it holds >= 2 nonzeros per Bt row of its span
#define SBLAS_WIN_LAUNCH(RPW, W, MW)                                                                                 \
    do {                                                                                                             \
        constexpr int R = 16 * RPW;                                                                                  \
        const int np = (rows + R - 1) / R;                                                                           \
        dim3 grid((unsigned)np, (unsigned)(ldbt / 64));                                                              \
        /* per device and cheap: set on every launch (one process may drive several GPUs) */                        \
        (void)hipFuncSetAttribute((const void *)spmm_window_kernel<RPW, W, MW>,                                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_lds_bytes(W));                \
        hipLaunchKernelGGL((spmm_window_kernel<RPW, W, MW>), grid, dim3(WIN_THREADS), win_lds_bytes(W), s, rows,     \
                           cols, np, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, dens);                   \
    } while (0)
            switch (variant) {
            case SPMM_VARIANT_WINDOW_R32: SBLAS_WIN_LAUNCH(2, 64, 8); break;   // 64 KiB LDS, 2 blocks/CU
            case SPMM_VARIANT_WINDOW_R128: SBLAS_WIN_LAUNCH(8, 128, 4); break; // 128 KiB LDS, 1 block/CU
            case SPMM_VARIANT_WINDOW_R64W64: SBLAS_WIN_LAUNCH(4, 64, 8); break;
            case SPMM_VARIANT_WINDOW_R32W128: SBLAS_WIN_LAUNCH(2, 128, 4); break;
            default: SBLAS_WIN_LAUNCH(4, 128, 4); break;
            }
#undef SBLAS_WIN_LAUNCH
        }
    } else if (ldbt == 32) {
        hipLaunchKernelGGL(spmm_rowpanel_narrow_kernel<32>, dim3(panels), dim3(256), 0, s, rows, rowptr, colidx,
                           val, Bt, n, alpha, beta, C, ldc);
    } else if (ldbt == 16) {
        hipLaunchKernelGGL(spmm_rowpanel_narrow_kernel<16>, dim3(panels), dim3(256), 0, s, rows, rowptr, colidx,
                           val, Bt, n, alpha, beta, C, ldc);
    } else {
        // n <= 8.  Long rows: a wave per row, eight sums per lane.  SBLAS_SPMM_VARIANT=direct keeps the lane-group kernel.
        const double avg = rows > 0 ? (double)nnz / (double)rows : 0.0;
        const char *ra = getenv("SBLAS_ROWS8_MIN_AVG"); /* experiments: row length from which a wave owns a row */
        // (banded-random rows, band +-20000, 600 k rows, N = 8: 64 / 128 / 200 / 300 per row: the lane groups win by
        //  25 / 30 / 2 / 0 %; bench matrix, 399 per row, band +-2000: the wave per row wins by 20 % -- tools/rows8_threshold.py)
        if (avg >= (ra ? atof(ra) : 256.0) && variant != SPMM_VARIANT_DIRECT)
            hipLaunchKernelGGL(spmm_rows8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, rows, cols, rowptr,
                               colidx, val, Bt, n, alpha, beta, C, ldc);
        else
            hipLaunchKernelGGL(spmm_rowpanel_narrow_kernel<8>, dim3(panels), dim3(256), 0, s, rows, rowptr, colidx,
                               val, Bt, n, alpha, beta, C, ldc);
    }
    return hipGetLastError();
}

hipError_t prof_stats(unsigned long long out[16], bool reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z);
    }
    return e;
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
    // Rows up to 96 nonzeros on average: the generic lanes-per-row kernel.  SBLAS_SPMV_VARIANT=burst|window|flat2|
    // flat4|flat8 select experimental long-row kernels (all correct, all slower) for A/B runs and tests.
    const char *sv = getenv("SBLAS_SPMV_VARIANT");
    if (avg > 96.0 && sv && (!strcmp(sv, "flat4") || !strcmp(sv, "flat8") || !strcmp(sv, "flat2"))) {
        const dim3 g((unsigned)((rows + 3) / 4));
        if (!strcmp(sv, "flat8"))
            hipLaunchKernelGGL(spmv_csr_flat_kernel<8>, g, dim3(256), 0, s, rows, rowptr, colidx, val, x, alpha, beta, y);
        else if (!strcmp(sv, "flat4"))
            hipLaunchKernelGGL(spmv_csr_flat_kernel<4>, g, dim3(256), 0, s, rows, rowptr, colidx, val, x, alpha, beta, y);
        else
            hipLaunchKernelGGL(spmv_csr_flat_kernel<2>, g, dim3(256), 0, s, rows, rowptr, colidx, val, x, alpha, beta, y);
        return hipGetLastError();
    }
#define SBLAS_SPMV_LDS(RWV, SV)                                                                                      \
    do {                                                                                                             \
        (void)hipFuncSetAttribute((const void *)spmv_csr_lds_kernel<RWV, SV>,                                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)((SPMV_LDS_CAP + 128) * sizeof(double))); \
        hipLaunchKernelGGL((spmv_csr_lds_kernel<RWV, SV>),                                                           \
                           dim3((unsigned)((rows + SPMV_LDS_ROWS * RWV - 1) / (SPMV_LDS_ROWS * RWV))), dim3(1024),    \
                           (SPMV_LDS_CAP + 128) * sizeof(double), s, rows, cols, rowptr, colidx, val, x, alpha, beta, y); \
        return hipGetLastError();                                                                                    \
    } while (0)
    // long rows: x window in LDS (bench matrix: 70-73 us vs 82-85 us for the lanes-per-row kernel); a block whose
    // rows span more than the LDS window degrades to global gathers by itself.  SBLAS_SPMV_VARIANT=plain keeps the
    // lanes-per-row kernel for A/B runs.
    // slices in flight per row: ~1.3-1.5 x the row length in 64-lane slices (600 k banded rows of 100 / 130 / 160 /
    // 200 / 260, band +-2000: S = 2 / 3 / 3 / 4 / 7 take 273 / 307 / 315 / 360 / 433 us against 329 / 337 / 345 / 360 /
    // 451 us with S = 4 throughout; the same order on a +-20000 band, tools/spmv_rowlen_sweep.py)
    if (avg > 96.0 && (!sv || !*sv || !strcmp(sv, "auto"))) {
        if (avg <= 115.0) SBLAS_SPMV_LDS(1, 2);
        if (avg <= 180.0) SBLAS_SPMV_LDS(1, 3);
        if (avg <= 230.0) SBLAS_SPMV_LDS(1, 4);
        SBLAS_SPMV_LDS(1, 7);
    }
    if (avg > 96.0 && sv && !strcmp(sv, "lds")) SBLAS_SPMV_LDS(1, 7);
    if (avg > 96.0 && sv && !strcmp(sv, "lds2")) SBLAS_SPMV_LDS(2, 7);
    if (avg > 96.0 && sv && !strcmp(sv, "lds2s4")) SBLAS_SPMV_LDS(2, 4);
    if (avg > 96.0 && sv && !strcmp(sv, "lds1s4")) SBLAS_SPMV_LDS(1, 4);
    if (avg > 32.0 && sv && !strcmp(sv, "lds1s2")) SBLAS_SPMV_LDS(1, 2);
    if (avg > 96.0 && sv && !strcmp(sv, "lds1s3")) SBLAS_SPMV_LDS(1, 3);
#define SBLAS_SPMV_SEG(RV, SV)                                                                                       \
    do {                                                                                                             \
        hipLaunchKernelGGL((spmv_csr_seg_kernel<RV, SV>), dim3((unsigned)((rows + 4 * RV - 1) / (4 * RV))), dim3(256), \
                           0, s, rows, rowptr, colidx, val, x, alpha, beta, y);                                      \
        return hipGetLastError();                                                                                    \
    } while (0)
    // medium rows: R rows per wave, segmented (Queen-like rows, 73 per row: 232 us vs 395 us; banded synthetic rows of
    // 36 / 72 / 90: 122 / 266 / 351 us vs 150 / 339 / 375 us for the lanes-per-row kernel; below ~32 per row the
    // lanes-per-row kernel wins)
    if (!sv || !*sv || !strcmp(sv, "auto")) {
        if (avg > 48.0 && avg <= 96.0) SBLAS_SPMV_SEG(4, 5);
    }
    // short and medium rows (5 < avg <= 48): 256 rows per block streamed through LDS, in runs of up to 6144 products
    // (stencil-like rows of 7 / 13 / 27: 108 / 177 / 344 us vs 143 / 277 / 498 us for the lanes-per-row and segmented
    // kernels; banded-random rows of 14 / 20 / 28 / 36 / 48: 46 / 62 / 85 / 116 / 161 vs 49 / 71 / 94 / 128 / 194).
    // Above 48 the segmented kernel stays (Queen-like rows of 69: 52 us vs 85 us for the stream form, which needs
    // three runs per block there); at 5 and below the lanes-per-row kernel is as fast or faster.
    if ((sv && !strcmp(sv, "stream")) || ((!sv || !*sv || !strcmp(sv, "auto")) && avg > 5.0 && avg <= 48.0)) {
        hipLaunchKernelGGL(spmv_csr_stream_kernel, dim3((unsigned)((rows + ST_ROWS - 1) / ST_ROWS)), dim3(ST_ROWS), 0, s, rows,
                           rowptr, colidx, val, x, alpha, beta, y);
        return hipGetLastError();
    }
    if (sv && !strcmp(sv, "seg4")) SBLAS_SPMV_SEG(4, 5);
    if (sv && !strcmp(sv, "seg3")) SBLAS_SPMV_SEG(3, 4);
    if (sv && !strcmp(sv, "seg8")) SBLAS_SPMV_SEG(8, 5);
    if (sv && !strcmp(sv, "seg2")) SBLAS_SPMV_SEG(2, 3);
    if (sv && !strcmp(sv, "nogather")) { // diagnostics only
        hipLaunchKernelGGL((spmv_csr_kernel<64, false>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, rows, rowptr,
                           colidx, val, x, alpha, beta, y);
        return hipGetLastError();
    }
    if (avg > 96.0 && sv && !strcmp(sv, "burst")) {
        hipLaunchKernelGGL(spmv_csr_burst_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, rows, rowptr,
                           colidx, val, x, alpha, beta, y);
        return hipGetLastError();
    }
    if (avg > 48.0 && sv && !strcmp(sv, "window")) {
        hipLaunchKernelGGL(spmv_csr_window_kernel, dim3((unsigned)((rows + SPMV_BLOCK_ROWS - 1) / SPMV_BLOCK_ROWS)),
                           dim3(512), SPMV_WINDOW_CAP * sizeof(double), s, rows, cols, rowptr, colidx, val, x, alpha,
                           beta, y);
        return hipGetLastError();
    }
    if (avg <= 6.0) return spmv_go<4>(s, rows, rowptr, colidx, val, x, alpha, beta, y);
    if (avg <= 12.0) return spmv_go<8>(s, rows, rowptr, colidx, val, x, alpha, beta, y);
    if (avg <= 24.0) return spmv_go<16>(s, rows, rowptr, colidx, val, x, alpha, beta, y);
    if (avg <= 48.0) return spmv_go<32>(s, rows, rowptr, colidx, val, x, alpha, beta, y);
    return spmv_go<64>(s, rows, rowptr, colidx, val, x, alpha, beta, y);
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

hipError_t launch_sum_replicas(hipStream_t s, const ReplicaPtrs &bufs, int g, int64_t n)
{
    hipLaunchKernelGGL(sum_replicas_kernel, dim3(capped_grid(n, 256)), dim3(256), 0, s, bufs, g, n);
    return hipGetLastError();
}

} // namespace sblas
