// scratch microbenchmark (not product code): what does one (row, tile) visit of the streaming SpMM kernels cost,
// component by component?  One 1024-thread workgroup per CU (128 KiB of LDS), NW waves run the visit loop, the
// others only take part in the barriers.  MODE bits:
//   1  select (7 VALU + 2 SALU, written out as in the product kernel)      2  sixteen DPP slots (8 ds_read_b128, 16 fmac)
//   4  count / validate / cursor update / rare-branch test (scalar)        8  window load (two buffer loads, L2-resident)
//  16  tile DMA (64 KiB per tile, L2-resident)                             32  barrier per tile
// Prints ns per tile and cycles per visit-round at the clock measured with GRBM-free arithmetic (2.1 GHz assumed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../s-blas_amd/csrc/kernels.hip"

using namespace sblas;

template <int MODE>
__global__ __launch_bounds__(1024) void visit_loop(const int *__restrict__ colidx, const double *__restrict__ val,
                                                   const double *__restrict__ Bt, int tiles, int nw, int nnz,
                                                   double *sink)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *zero_row = smem + 2 * W2_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    if (tid < 64) zero_row[tid] = 0.0;
    for (int i = tid; i < 2 * W2_TILE; i += 1024) smem[i] = 1.0;
    double acc[4][4];
    for (int r = 0; r < 4; ++r)
        for (int j = 0; j < 4; ++j) acc[r][j] = 0.0;
    const int eidx = ((lane & 15) << 2) + (lane >> 4);
    const int eload = eidx < 32 ? eidx : 0x40000000;
    const int wstart = (blockIdx.x * 16 + wave) * 4096 % (nnz - 65536);
    const sblas_rsrc_t rc = make_rsrc(colidx + wstart, 4u, (unsigned)(nnz - wstart));
    const sblas_rsrc_t rv = make_rsrc(val + wstart, 8u, (unsigned)(nnz - wstart));
    int cur[4] = {0, 400, 800, 1200};
    int end[4] = {400, 800, 1200, 1600};
    unsigned long long viol = 0;
    int wcb[4];
    double wvb[4];
    for (int r = 0; r < 4; ++r) {
        wcb[r] = (eidx * 11) & 127; // ascending-ish fake columns inside the tile
        wvb[r] = 1.0;
    }
    const unsigned ldb8 = 512;
    const unsigned pair_off = (unsigned)(lane >> 5) * ldb8 + (unsigned)((lane & 31) << 1) * 8u;
    const char *bt_bytes = reinterpret_cast<const char *>(Bt);
    if (MODE & 8)
        for (int r = 0; r < 3; ++r) window_issue(rc, rv, cur[r], eload, wcb[r], wvb[r]);
    __syncthreads();
    if (wave < nw) {
        for (int t = 0; t < tiles; ++t) {
            const int cb = t & 1;
            if (MODE & 16) {
                const int r0 = ((blockIdx.x * 5 + t) & 31) * 128 + wave * 8;
                const unsigned lds0 = (unsigned)(uintptr_t)(smem + (cb ^ 1) * W2_TILE) + (unsigned)wave * 4096u;
                const char *p = bt_bytes + (size_t)((unsigned)r0 * ldb8);
                if (wave < 16) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dma_rows_scalar(lds0 + i * 1024u, pair_off, p + (size_t)(2u * i) * ldb8);
                }
            }
            const int tile_lo = 0;
            const unsigned tile_base = (unsigned)(uintptr_t)(smem + cb * W2_TILE);
            const unsigned lb = tile_base + (unsigned)(lane & 15) * 16u;
            const unsigned zero_rel = (unsigned)(uintptr_t)zero_row - tile_base;
            auto visit = [&](auto rc_) {
                constexpr int r = decltype(rc_)::value;
                constexpr int rn = (r + 3) % 4;
                if (MODE & 8) {
                    window_issue(rc, rv, cur[rn], eload, wcb[rn], wvb[rn]);
                    if (MODE & 16) {
                        if (r == 3) asm volatile("s_waitcnt vmcnt(6)" : "+v"(wcb[r]), "+v"(wvb[r])::"memory");
                        else asm volatile("s_waitcnt vmcnt(10)" : "+v"(wcb[r]), "+v"(wvb[r])::"memory");
                    } else {
                        window_wait<6>(wcb[r], wvb[r]);
                    }
                    // keep the fake columns in the tile whatever was loaded
                    wcb[r] &= 127;
                }
                double &q0 = acc[r][0], &q1 = acc[r][1], &q2 = acc[r][2], &q3 = acc[r][3];
                const int cnt = min(32, end[r] - cur[r]);
                unsigned co = zero_rel;
                double gv = 0.0;
                unsigned long long m = 0xfffull;
                if (MODE & 1) window_select(wcb[r], wvb[r], tile_lo, cnt, eidx, zero_rel, co, gv, m);
                if (MODE & 2) { SBLAS_QSTEP4(0, 1, 2, 3); }
                if (MODE & 4) {
                    int take = mask_count(m);
                    take = min(take, 12);
                    viol |= m ^ __builtin_amdgcn_ballot_w64(eidx < take);
                    const bool more = take > 16 || (take >= cnt && cur[r] + take < end[r]);
                    if (__builtin_expect(!more, 1)) cur[r] += take;
                    else cur[r] = end[r] - 400; // never in this benchmark
                    if (cur[r] >= end[r] - 40) cur[r] = end[r] - 400;
                }
            };
            visit(std::integral_constant<int, 0>{});
            visit(std::integral_constant<int, 1>{});
            visit(std::integral_constant<int, 2>{});
            visit(std::integral_constant<int, 3>{});
            if (MODE & 32) __syncthreads();
        }
    } else if (MODE & 32) {
        for (int t = 0; t < tiles; ++t) __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double s = 0;
    for (int r = 0; r < 4; ++r)
        for (int j = 0; j < 4; ++j) s += acc[r][j];
    if (s == 12345.678 || viol == 77 || wcb[0] + wcb[1] + wcb[2] + wcb[3] == -5) sink[0] = s + cur[0] + cur[1] + cur[2] + cur[3];
}

template <int MODE> void run(const int *ci, const double *v, const double *bt, int nw, int nnz, double *sink)
{
    hipFuncSetAttribute((const void *)visit_loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES);
    const int tiles = 2000, blocks = 256;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    visit_loop<MODE><<<blocks, 1024, W2_LDS_BYTES>>>(ci, v, bt, 50, nw, nnz, sink);
    hipEventRecord(a);
    visit_loop<MODE><<<blocks, 1024, W2_LDS_BYTES>>>(ci, v, bt, tiles, nw, nnz, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double ns_tile = ms * 1e6 / tiles;
    printf("mode %2d%s%s%s%s%s%s  waves %2d: %7.1f ns/tile  = %6.0f clk per visit-round (2.1 GHz)\n", MODE,
           (MODE & 1) ? " sel" : "    ", (MODE & 2) ? " math" : "     ", (MODE & 4) ? " book" : "     ",
           (MODE & 8) ? " win" : "    ", (MODE & 16) ? " dma" : "    ", (MODE & 32) ? " bar" : "    ", nw, ns_tile,
           ns_tile * 2.1 / 4);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int nnz = 8 << 20;
    int *ci;
    double *v, *bt, *sink;
    hipMalloc(&ci, (size_t)nnz * 4);
    hipMalloc(&v, (size_t)nnz * 8);
    hipMalloc(&bt, (size_t)4200 * 64 * 8);
    hipMalloc(&sink, 8);
    hipMemset(ci, 0, (size_t)nnz * 4);
    hipMemset(v, 0, (size_t)nnz * 8);
    hipMemset(bt, 0, (size_t)4200 * 64 * 8);
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("start\n");
    for (int nw : {16, 12, 8}) {
        run<0>(ci, v, bt, nw, nnz, sink);
        run<1>(ci, v, bt, nw, nnz, sink);
        run<4>(ci, v, bt, nw, nnz, sink);
        run<5>(ci, v, bt, nw, nnz, sink);
        run<2>(ci, v, bt, nw, nnz, sink);
        run<3>(ci, v, bt, nw, nnz, sink);
        run<7>(ci, v, bt, nw, nnz, sink);
        run<8>(ci, v, bt, nw, nnz, sink);
        run<15>(ci, v, bt, nw, nnz, sink);
        run<16>(ci, v, bt, nw, nnz, sink);
        run<31>(ci, v, bt, nw, nnz, sink);
        run<32>(ci, v, bt, nw, nnz, sink);
        run<39>(ci, v, bt, nw, nnz, sink);
        run<47>(ci, v, bt, nw, nnz, sink);
        run<63>(ci, v, bt, nw, nnz, sink);
        printf("\n");
    }
    return 0;
}
