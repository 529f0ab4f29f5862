// spmm_mfma.hip -- SpMM on the fp64 matrix cores (v_mfma_f64_16x16x4_f64) for row panels whose nonzeros sit in dense
// sub-blocks (supernodal / multi-dof FEM matrices); the "MFMA only where a row panel is dense enough" part of the
// path that replaces cusparseSpMM (reference spmm.h:112-149).
//
// A wave owns 16 consecutive matrix rows and a tile of 16*NT dense columns (NT accumulators of 16 x 16).  It walks the
// union of its rows' columns in chunks of 64 columns -- the chunk that holds the smallest not yet consumed column of
// any of the 16 rows, so empty stretches cost nothing and nothing is assumed about the order inside a row:
//   1. every row's run of entries inside the chunk is scattered into a wave-private 16 x 64 fp64 image in LDS (laid
//      out as sixteen 16 x 4 operand blocks; LDS atomic adds, so duplicate entries sum) and a 64-bit mask records the
//      columns that hold an entry;
//   2. for every 4-column block with an entry: A operand = one ds_read_b64 per lane, B operand = 16-byte loads of the
//      four Bt rows (columns without an entry read the workspace's all-zero row instead, so a row of B that no
//      nonzero refers to cannot leak in), NT MFMAs;
//   3. the scattered positions are cleared again (the image stays all-zero between chunks).
// L2 -> CU traffic per visited block is 2 KiB (NT = 4) for up to 64 nonzeros, against 512 bytes per nonzero in the
// direct kernel; the price is arithmetic on the blocks' zeros -- fp64 MFMA has the same peak as fp64 vector FMA on
// gfx950, so this pays from about 40 % block fill (the classifier samples the fill per panel).
// An FMA cannot separate 0 * Inf from a product that belongs to the row: when stage 1 met a non-finite value in B,
// the panels classified for this kernel are computed by the vector kernels instead (flag in the workspace).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace sblas {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int MFMA_CHUNK = 64;                    // columns per chunk: sixteen 4-column operand blocks
constexpr int MFMA_IMG = 16 * MFMA_CHUNK;         // doubles per wave image (8 KiB)

template <int NT>
__global__ __launch_bounds__(MFMA_MAX_WAVES * 64) void spmm_mfma_kernel(
    int rows, int cols, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, const int2 *__restrict__ info, const int *__restrict__ tail,
    const int *__restrict__ cls, int panel_rows, int epoch, unsigned long long *__restrict__ stats)
{
    constexpr int NCOLS = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int panel = blockIdx.x;
    // no panel of this call was classified for the matrix cores: one scalar load and out
    if (tail[TAIL_MFMA_EPOCH] != epoch) return;
    // this kernel owns the panel when the classifier marked it and stage 1 saw only finite values in B
    const int c = cls[panel];
    if ((c != PANEL_MFMA_W && c != PANEL_MFMA_D) || tail[TAIL_NONFINITE] == tail[TAIL_STAGE_EPOCH]) return;
    (void)info;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int i = lane & 15, j = lane >> 4;
    const int row0 = panel * panel_rows;
    const int prow = min(panel_rows, rows - row0); // rows of this panel
    const int col0 = blockIdx.y * NCOLS;
    double *img = smem + wave * MFMA_IMG;
    unsigned long long *occ = reinterpret_cast<unsigned long long *>(smem + nwaves * MFMA_IMG) + wave;

    // the image starts all-zero
#pragma unroll
    for (int u = 0; u < MFMA_IMG / 64; ++u) img[u * 64 + lane] = 0.0;
    if (lane == 0) *occ = 0ull;

    const int lrow = wave * 16 + i; // row inside the panel
    int p = 0, pend = 0;
    if (lrow < prow) {
        p = rowptr[row0 + lrow];
        pend = rowptr[row0 + lrow + 1];
    }
    v4d acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};

    const unsigned ldb8 = (unsigned)ldbt * 8u;
    const char *bt_bytes = reinterpret_cast<const char *>(Bt);
    const unsigned lane_col = (unsigned)(col0 + 2 * i) * 8u; // this lane's two columns of every 32-column slice
    const unsigned zero_off = (unsigned)cols * ldb8;

    for (;;) {
        // the chunk that holds the smallest pending column of the 16 rows
        int h = (p < pend) ? colidx[p] : 0x7fffffff;
        h = min(h, __shfl_xor(h, 1, 64));
        h = min(h, __shfl_xor(h, 2, 64));
        h = min(h, __shfl_xor(h, 4, 64));
        h = min(h, __shfl_xor(h, 8, 64));
        const int hmin = __builtin_amdgcn_readfirstlane(h);
        if (hmin == 0x7fffffff) break;
        const int base = hmin & ~(MFMA_CHUNK - 1);
        // 1. scatter every row's run inside [base, base + 64): lanes j = 0..3 of row i take entries p + 4m + j
        int cnt = 0;
        bool open = true; // this row's run has not ended yet
        for (int m = 0;; ++m) {
            const int e = p + 4 * m + j;
            int c = -1;
            double v = 0.0;
            if (open && e < pend) {
                c = colidx[e];
                v = val[e];
            }
            const unsigned rel = (unsigned)(c - base);
            const bool in = open && e < pend && rel < (unsigned)MFMA_CHUNK;
            // a run is a PREFIX of the row's pending entries: lane j takes its entry only if lanes 0..j-1 did
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(in) >> i;
            const int b0 = (int)(mask & 1ull), b1 = (int)((mask >> 16) & 1ull), b2 = (int)((mask >> 32) & 1ull),
                      b3 = (int)((mask >> 48) & 1ull);
            const int t = b0 ? (b1 ? (b2 ? (b3 ? 4 : 3) : 2) : 1) : 0;
            if (j < t) {
                // image position: block (rel >> 2), row i, k (rel & 3)
                __hip_atomic_fetch_add(&img[((rel >> 2) * 16 + i) * 4 + (rel & 3u)], v, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(occ, 1ull << rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            cnt += t;
            open = open && t == 4;
            if (__builtin_amdgcn_ballot_w64(open) == 0ull) break;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): this wave's LDS atomics have landed (in-order per wave)
        const unsigned long long occ64 = *occ;
        // 2. MFMA over the blocks that hold an entry
        unsigned blocks = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) blocks |= ((occ64 >> (4 * b)) & 0xfull) ? (1u << b) : 0u;
        blocks = (unsigned)__builtin_amdgcn_readfirstlane((int)blocks);
        while (blocks) {
            const int b = __builtin_ctz(blocks);
            blocks &= blocks - 1;
            const double a = img[(b * 16 + i) * 4 + j];
            const bool used = (occ64 >> (4 * b + j)) & 1ull;
            const int brow = base + 4 * b + j;
            const unsigned roff = (used && brow < cols) ? (unsigned)brow * ldb8 : zero_off;
#pragma unroll
            for (int s = 0; s < NT / 2; ++s) {
                const double2 bb = *reinterpret_cast<const double2 *>(bt_bytes + (size_t)(roff + lane_col + (unsigned)s * 256u));
                acc[2 * s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb.x, acc[2 * s], 0, 0, 0);
                acc[2 * s + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb.y, acc[2 * s + 1], 0, 0, 0);
            }
        }
        // 3. clear the scattered positions and the mask (same entries, same addresses)
        for (int m = 0;; ++m) {
            const int q = 4 * m + j;
            if (q < cnt) {
                const unsigned rel = (unsigned)(colidx[p + q] - base);
                img[((rel >> 2) * 16 + i) * 4 + (rel & 3u)] = 0.0;
            }
            if (__builtin_amdgcn_ballot_w64(4 * (m + 1) < cnt) == 0ull) break;
        }
        if (lane == 0) *occ = 0ull;
        p += cnt;
    }
    if (tid == 0 && blockIdx.y == 0) atomicAdd(&stats[3], 1ull);

    // park the panel as [column][row] and write it back along rows (contiguous in column-major C)
    __syncthreads(); // every wave is done with its image
    double *ctile = smem;
    const int pr1 = panel_rows + 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int cc = 32 * (t >> 1) + 2 * i + (t & 1); // column inside the tile
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = wave * 16 + j + 4 * r;
            if (rr < panel_rows) ctile[cc * pr1 + rr] = acc[t][r];
        }
    }
    __syncthreads();
    const int ncols = min(NCOLS, n - col0);
    for (int idx = tid; idx < NCOLS * panel_rows; idx += blockDim.x) {
        const int r = idx % panel_rows, cj = idx / panel_rows;
        if (r < prow && cj < ncols) {
            double *dst = C + (int64_t)(col0 + cj) * ldc + (row0 + r);
            const double sres = alpha * ctile[cj * pr1 + r];
            *dst = (beta == 0.0) ? sres : fma(beta, *dst, sres);
        }
    }
}

hipError_t launch_spmm_mfma(hipStream_t s, int rows, int cols, const int *rowptr, const int *colidx, const double *val,
                            const double *Bt, int64_t ldbt, int n, double alpha, double beta, double *C, int64_t ldc,
                            const int2 *info, const int *tail, const int *cls, int panel_rows, int npanels, int epoch,
                            unsigned long long *stats)
{
    const int waves = (panel_rows + 15) / 16;
    if (waves < 1 || waves > MFMA_MAX_WAVES) return hipErrorInvalidValue;
    const int ncols = ldbt == 64 ? 64 : 128;
    const size_t img_bytes = (size_t)waves * MFMA_IMG * sizeof(double) + (size_t)waves * 8 + 64;
    const size_t ctile_bytes = (size_t)ncols * (size_t)(panel_rows + 1) * sizeof(double);
    const size_t lds = img_bytes > ctile_bytes ? img_bytes : ctile_bytes;
    if (ldbt == 64) {
        raise_dynamic_lds((const void *)spmm_mfma_kernel<4>, lds);
        hipLaunchKernelGGL(spmm_mfma_kernel<4>, dim3((unsigned)npanels, 1u), dim3((unsigned)waves * 64u), lds, s, rows, cols,
                           rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, info, tail, cls, panel_rows, epoch, stats);
    } else {
        raise_dynamic_lds((const void *)spmm_mfma_kernel<8>, lds);
        hipLaunchKernelGGL(spmm_mfma_kernel<8>, dim3((unsigned)npanels, (unsigned)(ldbt / 128)), dim3((unsigned)waves * 64u),
                           lds, s, rows, cols, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, info, tail,
                           cls, panel_rows, epoch, stats);
    }
    return hipGetLastError();
}

} // namespace sblas
