// spmm_mfma.hip -- SpMM on the fp64 matrix cores (v_mfma_f64_16x16x4_f64) for row panels whose nonzeros sit in dense
// sub-blocks (supernodal / multi-dof FEM matrices); the "MFMA only where a row panel is dense enough" part of the
// path that replaces cusparseSpMM (reference spmm.h:112-149).
//
// A wave owns 16 consecutive matrix rows and a tile of 16*NT dense columns (NT accumulators of 16 x 16).  It walks the
// union of its rows' columns in chunks of 64 columns -- the chunk that holds the smallest not yet consumed column of
// any of the 16 rows, so empty stretches cost nothing and nothing is assumed about the order inside a row:
//   1. every row's run of entries inside the chunk (a prefix of its 16-entry window, which is fetched one step ahead;
//      the four lanes of a quad serve one row) is scattered into a wave-private 16 x 64 fp64 image in LDS (laid out
//      as sixteen 16 x 4 operand blocks; LDS atomic adds, so duplicate entries sum) and one flag byte per chunk
//      column records the columns that hold an entry;
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
constexpr int MFMA_WAVE_LDS = MFMA_IMG * 8 + 64;  // bytes of LDS per wave: image + one flag byte per chunk column
constexpr int MFMA_WIN = 4;                       // window entries per lane: 16 per row

// Workgroups are dealt round-robin over the 8 XCDs: give every XCD one contiguous range of panels, so that the
// neighbouring panels -- which read overlapping rows of Bt -- share an L2 (speed only; bijective for any count).
__device__ __forceinline__ int mfma_xcd_panel(int b, int npanels)
{
    const int xcd = b & 7, idx = b >> 3;
    const int q = npanels >> 3, r = npanels & 7;
    return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
// min over the wave of a value that is equal in the four lanes of every quad
__device__ __forceinline__ int wave_min_quads(int x)
{
    constexpr int BIG = 0x7fffffff;
    x = min(x, __builtin_amdgcn_update_dpp(BIG, x, 0x114, 0xf, 0xf, false)); // row_shr:4
    x = min(x, __builtin_amdgcn_update_dpp(BIG, x, 0x118, 0xf, 0xf, false)); // row_shr:8 -> lanes 12..15 of a row
    return min(min(__builtin_amdgcn_readlane(x, 15), __builtin_amdgcn_readlane(x, 31)),
               min(__builtin_amdgcn_readlane(x, 47), __builtin_amdgcn_readlane(x, 63)));
}

template <int NT, int BATCH>
__global__ __launch_bounds__(MFMA_MAX_WAVES * 64) void spmm_mfma_kernel(
    int rows, int cols, const int *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ Bt, int64_t ldbt, int n, double alpha, double beta,
    double *__restrict__ C, int64_t ldc, const int2 *__restrict__ info, const int *__restrict__ tail,
    const int *__restrict__ cls, int panel_rows, int epoch, unsigned long long *__restrict__ stats)
{
    constexpr int NCOLS = 16 * NT;
    // BATCH: operand blocks fetched together (BATCH * NT / 2 16-byte loads in flight per lane and stage)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // no panel of this call was classified for the matrix cores: one scalar load and out
    if (tail[TAIL_MFMA_EPOCH] != epoch) return;
    const int panel = mfma_xcd_panel(blockIdx.x, gridDim.x);
    // this kernel owns the panel when the classifier marked it and stage 1 saw only finite values in B
    const int pc = cls[panel] & PANEL_CLASS_MASK;
    if ((pc != PANEL_MFMA_W && pc != PANEL_MFMA_D) || tail[TAIL_NONFINITE] == tail[TAIL_STAGE_EPOCH]) return;
    (void)info;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane >> 2, qj = lane & 3;  // entry phase: the four lanes of a quad share matrix row qi
    const int mi = lane & 15, mk = lane >> 4; // MFMA operands: A[row mi][k mk], B[k mk][column mi]
    const int row0 = panel * panel_rows;
    const int prow = min(panel_rows, rows - row0); // rows of this panel
    const int col0 = blockIdx.y * NCOLS;
    double *img = reinterpret_cast<double *>(smem_raw + wave * MFMA_WAVE_LDS);
    unsigned char *flags = reinterpret_cast<unsigned char *>(img + MFMA_IMG);

    // the image and the flags start all-zero
#pragma unroll
    for (int u = 0; u < MFMA_IMG / 64; ++u) img[u * 64 + lane] = 0.0;
    flags[lane] = 0;

    int p = 0, pend = 0;
    if (wave * 16 + qi < prow) {
        p = rowptr[row0 + wave * 16 + qi];
        pend = rowptr[row0 + wave * 16 + qi + 1];
    }
    v4d acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};

    const unsigned ldb8 = (unsigned)ldbt * 8u;
    const char *bt_bytes = reinterpret_cast<const char *>(Bt);
    const unsigned lane_col = (unsigned)(col0 + 2 * mi) * 8u; // this lane's two columns of every 32-column slice
    const unsigned zero_off = (unsigned)cols * ldb8;

    // window: entries p + qj + 4 u of row qi (16 per row), fetched one step ahead.  Unpredicated loads (clamped
    // index; the launch has at least one nonzero, so index 0 exists): hipcc keeps counted vmcnt waits only across
    // straight-line loads, and the pipeline below lives on them.
    int wc[MFMA_WIN];
    double wv[MFMA_WIN];
    auto load_window = [&]() {
#pragma unroll
        for (int u = 0; u < MFMA_WIN; ++u) {
            const int e = p + qj + 4 * u;
            const int idx = e < pend ? e : max(pend - 1, 0);
            const int c = colidx[idx];
            const double v = val[idx];
            wc[u] = e < pend ? c : 0x7fffffff;
            wv[u] = v;
        }
    };
    load_window();

    // One step of the walk = one 64-column chunk.  Two-stage software pipeline: a step scatters its entries, reads the
    // A operands of its (first BATCH) blocks and ISSUES their B loads; the MFMAs of those blocks run one step later,
    // after the next step's scatter, so that the B rows' trip from L2 / the Infinity Cache overlaps that work (and,
    // across the waves of a SIMD, the other waves' MFMAs).
    struct Stage {
        int nb;                       // blocks in flight (wave-uniform)
        double a[BATCH];
        double2 bb[BATCH][NT / 2];
    };
    auto run_mfma = [&](const Stage &st) {
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            if (q < st.nb) {
#pragma unroll
                for (int s = 0; s < NT / 2; ++s) {
                    acc[2 * s] = __builtin_amdgcn_mfma_f64_16x16x4f64(st.a[q], st.bb[q][s].x, acc[2 * s], 0, 0, 0);
                    acc[2 * s + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(st.a[q], st.bb[q][s].y, acc[2 * s + 1], 0, 0, 0);
                }
            }
        }
    };
    // operands of up to BATCH blocks of mask m (bit 4 b: block b) -> st; returns the remaining mask
    auto fetch_blocks = [&](Stage &st, unsigned long long m, unsigned long long occ64, int base) {
        st.nb = min(BATCH, (int)__builtin_popcountll(m));
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            const int b = m ? (int)(__builtin_ctzll(m) >> 2) : 0; // (absent slot: block 0 of the all-zero row, never used)
            const bool have = m != 0ull;
            m &= m - 1;
            st.a[q] = img[b * 64 + mi * 4 + mk];
            const int brow = base + 4 * b + mk;
            const bool live = have && ((occ64 >> (4 * b + mk)) & 1ull) && brow < cols;
            const unsigned roff = live ? (unsigned)brow * ldb8 : zero_off;
#pragma unroll
            for (int s = 0; s < NT / 2; ++s)
                st.bb[q][s] = *reinterpret_cast<const double2 *>(bt_bytes + (size_t)(roff + lane_col + (unsigned)s * 256u));
        }
        return m;
    };
    // returns false when the walk is over
    auto step = [&](Stage &cur, const Stage &prev) -> bool {
        // the chunk that holds the smallest pending column of the 16 rows (a row's head = its window entry 0)
        const int head = __builtin_amdgcn_update_dpp(0x7fffffff, wc[0], 0x00, 0xf, 0xf, false); // quad_perm:[0,0,0,0]
        const int hmin = wave_min_quads(head);
        if (hmin == 0x7fffffff) return false;
        const int base = hmin & ~(MFMA_CHUNK - 1);
        // a row's run inside [base, base + 64) is a PREFIX of its window (entry order 4 u + qj)
        unsigned rel[MFMA_WIN];
        int t = 0;
        bool open = true;
#pragma unroll
        for (int u = 0; u < MFMA_WIN; ++u) {
            rel[u] = (unsigned)(wc[u] - base);
            const bool in = rel[u] < (unsigned)MFMA_CHUNK; // (absent entries carry 0x7fffffff: never inside)
            const unsigned nib = (unsigned)(__builtin_amdgcn_ballot_w64(in) >> (lane & 60)) & 15u;
            const int tu = __builtin_ctz(~nib);          // leading run of the quad's four entries (0..4)
            t += open ? tu : 0;
            open = open && tu == 4;
        }
        // scatter the run into the image (atomic adds: duplicate entries sum), flag its columns
        bool taken[MFMA_WIN];
#pragma unroll
        for (int u = 0; u < MFMA_WIN; ++u) {
            taken[u] = 4 * u + qj < t;
            if (taken[u]) {
                __hip_atomic_fetch_add(&img[(rel[u] >> 2) * 64 + qi * 4 + (rel[u] & 3u)], wv[u], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                flags[rel[u]] = 1;
            }
        }
        // next step's window (a row whose window was used up entirely comes back to the same chunk if it has more)
        p += t;
        load_window();
        // columns that hold an entry; blocks that hold one (bit 4 b)
        const unsigned long long occ64 = __builtin_amdgcn_ballot_w64(flags[lane] != 0);
        unsigned long long m = occ64 | (occ64 >> 1);
        m |= m >> 2;
        m &= 0x1111111111111111ull;
        m = fetch_blocks(cur, m, occ64, base);
        // the previous step's blocks, whose B rows have had this step's scatter to arrive
        run_mfma(prev);
        // a chunk with more than BATCH blocks: the rest right away (the image must stay until all operands are read)
        while (m) {
            run_mfma(cur);
            m = fetch_blocks(cur, m, occ64, base);
        }
        // clear the scattered positions and the flags again
#pragma unroll
        for (int u = 0; u < MFMA_WIN; ++u)
            if (taken[u]) img[(rel[u] >> 2) * 64 + qi * 4 + (rel[u] & 3u)] = 0.0;
        if (lane < 8) reinterpret_cast<unsigned long long *>(flags)[lane] = 0ull;
        return true;
    };
    {
        Stage s0, s1;
        s0.nb = 0;
        s1.nb = 0;
        for (;;) {
            if (!step(s0, s1)) {
                run_mfma(s1);
                break;
            }
            if (!step(s1, s0)) {
                run_mfma(s0);
                break;
            }
        }
    }
    if (tid == 0 && blockIdx.y == 0) atomicAdd(&stats[3], 1ull);

    // park the panel as [column][row] in LDS and write it back along rows: a column of C gets the panel's rows as one
    // contiguous run (768 bytes for 96 rows; wave-private 128-byte pieces ran 6 % slower at N = 128).  alpha / beta are
    // applied here, so C is read and written exactly once.
    __syncthreads(); // every wave is done with its image
    double *ctile = reinterpret_cast<double *>(smem_raw);
    const int pr1 = panel_rows + 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int cc = 32 * (t >> 1) + 2 * mi + (t & 1); // column inside the tile
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = wave * 16 + mk + 4 * r;
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
    const size_t img_bytes = (size_t)waves * MFMA_WAVE_LDS;
    const size_t ctile_bytes = (size_t)(ldbt == 64 ? 64 : 128) * (size_t)(panel_rows + 1) * sizeof(double);
    size_t lds = img_bytes > ctile_bytes ? img_bytes : ctile_bytes;
    if ((size_t)options().tune[0] > lds && options().tune[0] <= 160 * 1024) lds = (size_t)options().tune[0]; // experiments: occupancy
#define SBLAS_MFMA_LAUNCH(NTV, BV, GY)                                                                                \
    do {                                                                                                             \
        raise_dynamic_lds((const void *)spmm_mfma_kernel<NTV, BV>, lds);                                             \
        hipLaunchKernelGGL((spmm_mfma_kernel<NTV, BV>), dim3((unsigned)npanels, (unsigned)(GY)),                     \
                           dim3((unsigned)waves * 64u), lds, s, rows, cols, rowptr, colidx, val, Bt, ldbt, n, alpha,  \
                           beta, C, ldc, info, tail, cls, panel_rows, epoch, stats);                                 \
    } while (0)
    const int batch = options().tune[1]; // experiments: operand blocks per stage
    if (ldbt == 64) {
        if (batch == 4) SBLAS_MFMA_LAUNCH(4, 4, 1);
        else if (batch == 3) SBLAS_MFMA_LAUNCH(4, 3, 1);
        else SBLAS_MFMA_LAUNCH(4, 2, 1);
    } else { // (256 columns per wave -- half the chunk steps of two 128-column passes -- spills: 128 accumulator registers)
        if (batch == 1) SBLAS_MFMA_LAUNCH(8, 1, ldbt / 128);
        else SBLAS_MFMA_LAUNCH(8, 2, ldbt / 128);
    }
#undef SBLAS_MFMA_LAUNCH
    return hipGetLastError();
}

} // namespace sblas
