// B-stationary-in-VGPRs microbenchmark ("next" 3): is there an operand path for the headline SpMM shape
// that does not read 512 bytes of LDS per nonzero?
//
// A wave keeps a slab of S = 96 rows of Bt x 64 columns in registers (lane = column, row r = the register pair
// v[16 + 2r : 17 + 2r]); the entries of A arrive through the SCALAR path (s_load_dwordx*: 12 column numbers and 12 values
// per (row, slab) visit, prefetched one visit ahead into a second set of SGPRs); per nonzero the wave executes
//     s_set_gpr_idx_on  s_col, 1            ; M0[7:0] = register offset of the B row (2 x relative column)
//     v_fma_f64  acc, v[16:17](+M0), s_val, acc
// -- no LDS and no vector-memory traffic per nonzero -- and per visit (ten nonzeros of one matrix row) one ds_add_f64
// of the 64 partial sums into the workgroup's C tile.  Variants: scalar instructions per nonzero in front of the index
// write (what a real kernel needs to turn a CSR column number into a register offset: a shift; a subtraction and a
// shift), s_set_gpr_idx_idx instead of s_set_gpr_idx_on, the FMAs alone; one or two waves per SIMD.
// Go / no-go (VERDICT): <= 8 clk per nonzero per SIMD, sustained, on L2-resident data; the LDS-tiled kernel runs at ~16.
//
//   hipcc -O3 --offload-arch=gfx950 tools/bstat_bench.hip -o tools/bstat_bench && tools/bstat_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define VISIT_BYTES 256 // 16 x int32 column slots (64 B) + 16 x fp64 value slots (128 B) + padding

#define BSTAT_CLOBBERS \
    "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v16", "v17", "v18", "v19", "v20", "v21", \
    "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", \
    "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", \
    "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", \
    "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", \
    "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", \
    "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", \
    "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", \
    "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", \
    "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", \
    "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", \
    "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", \
    "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", \
    "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", \
    "v204", "v205", "v206", "v207", \
    "s2", "s3", "s4", "s5", "s6", "s7", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", \
    "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", \
    "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", \
    "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", \
    "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", \
    "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", \
    "m0", "scc", "vcc", "memory"

#define STR2(x) #x
#define STR(x) STR2(x)
// one nonzero: column in s[C], value in s[V:V+1]
#define IDX_ON(C) "s_set_gpr_idx_on s[" STR(C) "], 0x1\n\t"
#define IDX_IDX(C) "s_set_gpr_idx_idx s[" STR(C) "]\n\t"
#define MUL(V, ACC) "v_mul_f64 " ACC ", v[16:17], s[" STR(V) ":" STR(V) "+1]\n\t"
#define FMA(V, ACC) "v_fma_f64 " ACC ", v[16:17], s[" STR(V) ":" STR(V) "+1], " ACC "\n\t"
#define SHL(C) "s_lshl_b32 s[" STR(C) "], s[" STR(C) "], 1\n\t"
#define SUB(C) "s_sub_u32 s[" STR(C) "], s[" STR(C) "], s7\n\t"
// mode 0: index write + FMA (the stream carries ready-made register offsets)
#define E0_FIRST(C, V, ACC) IDX_ON(C) MUL(V, ACC)
#define E0(C, V, ACC) IDX_ON(C) FMA(V, ACC)
// mode 1: + a shift; mode 2: + a subtraction and a shift
#define E1_FIRST(C, V, ACC) SHL(C) IDX_ON(C) MUL(V, ACC)
#define E1(C, V, ACC) SHL(C) IDX_ON(C) FMA(V, ACC)
#define E2_FIRST(C, V, ACC) SUB(C) SHL(C) IDX_ON(C) MUL(V, ACC)
#define E2(C, V, ACC) SUB(C) SHL(C) IDX_ON(C) FMA(V, ACC)
// mode 3: mode 0 with s_set_gpr_idx_idx behind the first entry
#define E3(C, V, ACC) IDX_IDX(C) FMA(V, ACC)
// mode 4: the FMAs alone (fixed register): the floor of this loop
#define E4_FIRST(C, V, ACC) MUL(V, ACC)
#define E4(C, V, ACC) FMA(V, ACC)

// ten nonzeros of a visit: columns s[C0 .. C0+9], values s[V0 .. V0+19]
#define VISIT10(F, E, C0, V0, ACC)                                                                                     \
    F(C0, V0, ACC) E(C0 + 1, V0 + 2, ACC) E(C0 + 2, V0 + 4, ACC) E(C0 + 3, V0 + 6, ACC) E(C0 + 4, V0 + 8, ACC)           \
    E(C0 + 5, V0 + 10, ACC) E(C0 + 6, V0 + 12, ACC) E(C0 + 7, V0 + 14, ACC) E(C0 + 8, V0 + 16, ACC) E(C0 + 9, V0 + 18, ACC) \
    "s_set_gpr_idx_off\n\t"

// loads of one visit (12 columns, 12 values) at byte offset OFF of s[2:3]
#define LOADS(OFF, C0, V0)                                                                                             \
    "s_load_dwordx8 s[" STR(C0) ":" STR(C0) "+7], s[2:3], " STR(OFF) "\n\t"                                            \
    "s_load_dwordx4 s[" STR(C0) "+8:" STR(C0) "+11], s[2:3], " STR(OFF) "+0x20\n\t"                                    \
    "s_load_dwordx16 s[" STR(V0) ":" STR(V0) "+15], s[2:3], " STR(OFF) "+0x40\n\t"                                     \
    "s_load_dwordx8 s[" STR(V0) "+16:" STR(V0) "+23], s[2:3], " STR(OFF) "+0x80\n\t"

// SGPR map inside the block: s[2:3] stream pointer, s4 pairs of visits left, s5 visits left before the stream wraps,
//   s6 LDS row offset, s7 slab base; set X: columns s[16:27], values s[28:51]; set Y: columns s[52:63], values s[64:87]
#define BSTAT_KERNEL(NAME, F, E)                                                                                       \
    __global__ __launch_bounds__(512) void NAME(const char *stream, int visits_per_wave, int pairs, long long *cycles)  \
    {                                                                                                                 \
        extern __shared__ double ctile[]; /* 16 rows x 64 columns per wave */                                         \
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;                                                   \
        const int gw = blockIdx.x * (blockDim.x >> 6) + wave;                                                         \
        const char *mine = stream + (size_t)gw * visits_per_wave * VISIT_BYTES;                                       \
        for (int i = threadIdx.x; i < (int)(blockDim.x >> 6) * 16 * 64; i += blockDim.x) ctile[i] = 0.0;              \
        __syncthreads();                                                                                              \
        const unsigned lds = (unsigned)(uintptr_t)(ctile + wave * 16 * 64 + lane);                                    \
        const unsigned long long base = (unsigned long long)mine;                                                     \
        long long t0 = 0, t1 = 0;                                                                                     \
        asm volatile("s_mov_b64 s[2:3], %[base]\n\t"                                                                  \
                     "s_mov_b32 s4, %[pairs]\n\t"                                                                     \
                     "s_mov_b32 s5, %[vpw]\n\t"                                                                       \
                     "s_mov_b32 s6, 0\n\t"                                                                            \
                     "s_mov_b32 s7, 0\n\t" /* the slab: 96 rows x one double per lane, values in [1, 2) */            \
                     "v_mov_b32 v10, 0x3ff00000\n\t"                                                                  \
                     "v_lshlrev_b32 v11, 3, %[lane]\n\t"                                                              \
                     ".set r, 16\n\t"                                                                                 \
                     ".rept 96\n\t"                                                                                   \
                     "v_add_u32 v[r], r, v11\n\t"                                                                     \
                     "v_mov_b32 v[r+1], v10\n\t"                                                                      \
                     ".set r, r+2\n\t"                                                                                \
                     ".endr\n\t"                                                                                      \
                     "v_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"                    \
                     "s_memtime %[t0]\n\t" LOADS(0x0, 16, 28) "1:\n\t" /* ---- visit X */                             \
                     "s_waitcnt lgkmcnt(0)\n\t"                                                                       \
                     "v_add_u32 v9, s6, %[lds]\n\t"                                                                   \
                     "ds_add_f64 v9, v[6:7]\n\t" /* the previous visit's sums */                                      \
                     LOADS(0x100, 52, 64) VISIT10(F##_FIRST, E, 16, 28, "v[4:5]") /* ---- visit Y */                  \
                     "s_waitcnt lgkmcnt(0)\n\t"                                                                       \
                     "ds_add_f64 v9, v[4:5] offset:512\n\t"                                                           \
                     "s_add_u32 s2, s2, 0x200\n\t"                                                                    \
                     "s_addc_u32 s3, s3, 0\n\t"                                                                       \
                     "s_add_u32 s6, s6, 0x400\n\t"                                                                    \
                     "s_and_b32 s6, s6, 0x1fff\n\t" /* the stream is walked several times (L2-resident) */            \
                     "s_sub_u32 s5, s5, 2\n\t"                                                                        \
                     "s_cmp_gt_u32 s5, 1\n\t"                                                                         \
                     "s_cbranch_scc1 2f\n\t"                                                                          \
                     "s_mov_b64 s[2:3], %[base]\n\t"                                                                  \
                     "s_mov_b32 s5, %[vpw]\n\t"                                                                       \
                     "2:\n\t" LOADS(0x0, 16, 28) VISIT10(F##_FIRST, E, 52, 64, "v[6:7]")                              \
                     "s_sub_u32 s4, s4, 1\n\t"                                                                        \
                     "s_cmp_lg_u32 s4, 0\n\t"                                                                         \
                     "s_cbranch_scc1 1b\n\t"                                                                          \
                     "s_waitcnt lgkmcnt(0)\n\t"                                                                       \
                     "s_memtime %[t1]\n\t"                                                                            \
                     "s_waitcnt lgkmcnt(0)\n\t"                                                                       \
                     : [t0] "=&s"(t0), [t1] "=&s"(t1)                                                                 \
                     : [base] "s"(base), [pairs] "s"(pairs), [vpw] "s"(visits_per_wave), [lane] "v"(lane),            \
                       [lds] "v"(lds)                                                                                 \
                     : BSTAT_CLOBBERS);                                                                               \
        if (lane == 0) cycles[gw] = t1 - t0;                                                                          \
    }

BSTAT_KERNEL(bstat_mode0, E0, E0)
BSTAT_KERNEL(bstat_mode1, E1, E1)
BSTAT_KERNEL(bstat_mode2, E2, E2)
BSTAT_KERNEL(bstat_mode3, E0, E3)
BSTAT_KERNEL(bstat_mode4, E4, E4)


// ---- deeper prefetch: three SGPR sets of eight entries (8 columns + 16 value dwords), loads issued TWO visits ahead.
// Sets: X s[16:39], Y s[40:63], Z s[64:87] (columns first, then values).
#define LOADS8(OFF, C0)                                                                                                \
    "s_load_dwordx8 s[" STR(C0) ":" STR(C0) "+7], s[2:3], " STR(OFF) "\n\t"                                            \
    "s_load_dwordx16 s[" STR(C0) "+8:" STR(C0) "+23], s[2:3], " STR(OFF) "+0x40\n\t"
#define VISIT8(F, E, C0, ACC)                                                                                          \
    F(C0, C0 + 8, ACC) E(C0 + 1, C0 + 10, ACC) E(C0 + 2, C0 + 12, ACC) E(C0 + 3, C0 + 14, ACC) E(C0 + 4, C0 + 16, ACC)    \
    E(C0 + 5, C0 + 18, ACC) E(C0 + 6, C0 + 20, ACC) E(C0 + 7, C0 + 22, ACC) "s_set_gpr_idx_off\n\t"
#define BSTAT3_KERNEL(NAME, F, E)                                                                                      \
    __global__ __launch_bounds__(512) void NAME(const char *stream, int visits_per_wave, int triples, long long *cycles) \
    {                                                                                                                 \
        extern __shared__ double ctile[];                                                                             \
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;                   \
        const int gw = blockIdx.x * (blockDim.x >> 6) + wave;                                                         \
        const char *mine = stream + (size_t)gw * visits_per_wave * VISIT_BYTES;                                       \
        for (int i = threadIdx.x; i < (int)(blockDim.x >> 6) * 16 * 64; i += blockDim.x) ctile[i] = 0.0;              \
        __syncthreads();                                                                                              \
        const unsigned lds = (unsigned)(uintptr_t)(ctile + wave * 16 * 64 + lane);                                    \
        const unsigned long long base = (unsigned long long)mine;                                                     \
        long long t0 = 0, t1 = 0;                                                                                     \
        asm volatile("s_mov_b64 s[2:3], %[base]\n\t"                                                                  \
                     "s_mov_b32 s4, %[triples]\n\t"                                                                   \
                     "s_mov_b32 s5, %[vpw]\n\t"                                                                       \
                     "s_mov_b32 s6, 0\n\t"                                                                            \
                     "s_mov_b32 s7, 0\n\t"                                                                            \
                     "v_mov_b32 v10, 0x3ff00000\n\t"                                                                  \
                     "v_lshlrev_b32 v11, 3, %[lane]\n\t"                                                              \
                     ".set r, 16\n\t"                                                                                 \
                     ".rept 96\n\t"                                                                                   \
                     "v_add_u32 v[r], r, v11\n\t"                                                                     \
                     "v_mov_b32 v[r+1], v10\n\t"                                                                      \
                     ".set r, r+2\n\t"                                                                                \
                     ".endr\n\t"                                                                                      \
                     "v_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"                    \
                     "s_memtime %[t0]\n\t" LOADS8(0x0, 16) LOADS8(0x100, 40) "1:\n\t"                                  \
                     /* visit X: its loads were issued two visits ago; Y's (one visit ago) may still be in flight, but */ \
                     /* scalar loads return out of order, so the only safe wait is lgkmcnt(0) -- which is why the     */ \
                     /* loads of visit Z are issued BEHIND the wait, ahead of X's arithmetic                           */ \
                     "s_waitcnt lgkmcnt(0)\n\t"                                                                       \
                     "v_add_u32 v9, s6, %[lds]\n\t"                                                                   \
                     "ds_add_f64 v9, v[6:7]\n\t" LOADS8(0x200, 64) VISIT8(F##_FIRST, E, 16, "v[4:5]")                  \
                     "s_waitcnt lgkmcnt(0)\n\t"                                                                       \
                     "ds_add_f64 v9, v[4:5] offset:512\n\t"                                                           \
                     "s_add_u32 s2, s2, 0x300\n\t"                                                                    \
                     "s_addc_u32 s3, s3, 0\n\t"                                                                       \
                     "s_sub_u32 s5, s5, 3\n\t"                                                                        \
                     "s_cmp_gt_u32 s5, 2\n\t"                                                                         \
                     "s_cbranch_scc1 2f\n\t"                                                                          \
                     "s_mov_b64 s[2:3], %[base]\n\t"                                                                  \
                     "s_mov_b32 s5, %[vpw]\n\t"                                                                       \
                     "2:\n\t" LOADS8(0x0, 16) VISIT8(F##_FIRST, E, 40, "v[6:7]")                                       \
                     "s_waitcnt lgkmcnt(0)\n\t"                                                                       \
                     "ds_add_f64 v9, v[6:7] offset:1024\n\t"                                                          \
                     "s_add_u32 s6, s6, 0x600\n\t"                                                                    \
                     "s_and_b32 s6, s6, 0x1fff\n\t" LOADS8(0x100, 40) VISIT8(F##_FIRST, E, 64, "v[4:5]")               \
                     "v_mov_b32 v6, v4\n\tv_mov_b32 v7, v5\n\t"                                                        \
                     "s_sub_u32 s4, s4, 1\n\t"                                                                        \
                     "s_cmp_lg_u32 s4, 0\n\t"                                                                         \
                     "s_cbranch_scc1 1b\n\t"                                                                          \
                     "s_waitcnt lgkmcnt(0)\n\t"                                                                       \
                     "s_memtime %[t1]\n\t"                                                                            \
                     "s_waitcnt lgkmcnt(0)\n\t"                                                                       \
                     : [t0] "=&s"(t0), [t1] "=&s"(t1)                                                                 \
                     : [base] "s"(base), [triples] "s"(triples), [vpw] "s"(visits_per_wave), [lane] "v"(lane),        \
                       [lds] "v"(lds)                                                                                 \
                     : BSTAT_CLOBBERS);                                                                               \
        if (lane == 0) cycles[gw] = t1 - t0;                                                                          \
    }
BSTAT3_KERNEL(bstat3_mode0, E0, E0)
BSTAT3_KERNEL(bstat3_mode1, E1, E1)
BSTAT3_KERNEL(bstat3_mode4, E4, E4)

typedef void (*kern_t)(const char *, int, int, long long *);

int main()
{
    const int visits = 32;                  // per wave: 8 KB of stream, walked many times
    const int max_waves = 256 * 8 * 2;      // 2 workgroups of 8 waves per CU at most
    // two streams: slab-relative row numbers (modes 1, 2 turn them into register offsets themselves) and ready-made
    // register offsets 2 x row (modes 0, 3, 4)
    std::vector<char> h((size_t)max_waves * visits * VISIT_BYTES), h2(h.size());
    srand(211);
    for (size_t vtx = 0; vtx < (size_t)max_waves * visits; ++vtx) {
        int *c = reinterpret_cast<int *>(&h[vtx * VISIT_BYTES]);
        double *v = reinterpret_cast<double *>(&h[vtx * VISIT_BYTES + 64]);
        int *c2 = reinterpret_cast<int *>(&h2[vtx * VISIT_BYTES]);
        double *v2 = reinterpret_cast<double *>(&h2[vtx * VISIT_BYTES + 64]);
        int col = rand() % 8;
        for (int e = 0; e < 16; ++e) {
            c[e] = col % 96;
            c2[e] = 2 * (col % 96);
            v[e] = v2[e] = (double)rand() / RAND_MAX;
            col += 1 + rand() % 12;
        }
    }
    char *d, *d2;
    long long *cyc;
    hipMalloc(&d, h.size());
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    hipMalloc(&d2, h2.size());
    hipMemcpy(d2, h2.data(), h2.size(), hipMemcpyHostToDevice);
    hipMalloc(&cyc, max_waves * sizeof(long long));
    const kern_t kerns[5] = {bstat_mode0, bstat_mode1, bstat_mode2, bstat_mode3, bstat_mode4};
    const char *names[5] = {"idx_on + fma            ", "shift + idx_on + fma    ", "sub + shift + idx_on+fma", "idx_idx + fma           ",
                            "fma alone (no indexing) "};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int pairs = 4000; // 8000 visits = 80 000 nonzeros per wave
    std::vector<long long> hc(max_waves);
    printf("B-stationary SpMM inner loop, 96 x 64 slab of Bt in VGPRs, entries of A through s_load, ten per visit\n");
    for (int wps = 1; wps <= 2; ++wps) {           // waves per SIMD
        const int threads = 256 * wps;
        for (int m = 0; m < 5; ++m) {
            hipFuncSetAttribute((const void *)kerns[m], hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16 * 64 * 8);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kerns[m], dim3(256), dim3(threads), (size_t)(threads / 64) * 16 * 64 * 8, 0, (m == 1 || m == 2) ? d : d2, visits, pairs, cyc);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
            const int waves = 256 * threads / 64;
            hipMemcpy(hc.data(), cyc, waves * sizeof(long long), hipMemcpyDeviceToHost);
            double sum = 0;
            for (int i = 0; i < waves; ++i) sum += (double)hc[i];
            const double nnz_wave = 20.0 * pairs;
            const double clk_wave = sum / waves;   // s_memtime ticks = shader cycles
            printf("  %d wave(s)/SIMD  %s: %7.3f ms, %6.2f clk per nonzero per wave, %6.2f clk per nonzero per SIMD, clock %.2f GHz\n", wps,
                   names[m], ms, clk_wave / nnz_wave, clk_wave / nnz_wave / wps, clk_wave / (ms * 1e-3) / 1e9);
        }
    }
    // deeper prefetch: loads two visits ahead, eight nonzeros per visit
    const kern_t k3[3] = {bstat3_mode0, bstat3_mode1, bstat3_mode4};
    const char *n3[3] = {"idx_on + fma            ", "shift + idx_on + fma    ", "fma alone (no indexing) "};
    const int visits3 = 30, triples = 3000;
    printf("... the same with the scalar loads issued TWO visits ahead, eight nonzeros per visit\n");
    for (int wps = 1; wps <= 2; ++wps) {
        const int threads = 256 * wps;
        for (int m = 0; m < 3; ++m) {
            hipFuncSetAttribute((const void *)k3[m], hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16 * 64 * 8);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k3[m], dim3(256), dim3(threads), (size_t)(threads / 64) * 16 * 64 * 8, 0, m == 1 ? d : d2, visits3, triples, cyc);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
            const int waves = 256 * threads / 64;
            hipMemcpy(hc.data(), cyc, waves * sizeof(long long), hipMemcpyDeviceToHost);
            double sum = 0;
            for (int i = 0; i < waves; ++i) sum += (double)hc[i];
            const double nnz_wave = 24.0 * triples;
            const double clk_wave = sum / waves;
            printf("  %d wave(s)/SIMD  %s: %7.3f ms, %6.2f clk per nonzero per wave, %6.2f clk per nonzero per SIMD, clock %.2f GHz\n", wps,
                   n3[m], ms, clk_wave / nnz_wave, clk_wave / nnz_wave / wps, clk_wave / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
