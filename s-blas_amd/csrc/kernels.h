// kernels.h -- internal launcher interface between the kernel translation units and the C-ABI layer (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sblas {

constexpr int MAX_REPLICAS = 16;
struct ReplicaPtrs {
    double *p[MAX_REPLICAS];
};

// SpMM stage-2 selection (SBLAS_SPMM_VARIANT, experiments and tests only).  AUTO = panel classifier, then the
// LDS-tiled / MFMA kernels on the panels that qualify and a direct kernel on the rest.
enum {
    SPMM_VARIANT_AUTO = 0,
    SPMM_VARIANT_DIRECT_DPP = 1,  // "dpp":   every panel through the row-per-wave direct kernel
    SPMM_VARIANT_DIRECT_ROWS = 2, // "rows":  every panel through the four-rows-per-wave direct kernel
    SPMM_VARIANT_LANES = 3,       // "lanes": n <= 8 keeps the lane-group kernel whatever the row length (else AUTO)
    SPMM_VARIANT_DIRECT_MERGE = 4, // "merge": every panel through the three-rows-per-wave direct kernel (128-column tiles)
    SPMM_VARIANT_MFMA = 5,        // "mfma":  LDS-tiled panels through the MFMA kernel whatever their block density
    SPMM_VARIANT_NO_MFMA = 6,     // "nomfma": never the MFMA kernel
};
constexpr int SPMM_MIN_PANEL_ROWS = 32; // smallest classified panel (the workspace tail has room for rows / 32 panels)

// Workspace layout behind the staging copy Bt ((cols + 1) x ldbt doubles): TAIL_HDR ints, TAIL_PARTS int2 (partial
// column ranges of a row block), then one int2 (column span) per panel, then one int (class) per panel.
enum { TAIL_NONFINITE = 0,    // = TAIL_STAGE_EPOCH's value when the staging pass met an Inf / NaN in B
       TAIL_STAGE_EPOCH = 1,  // epoch of the staging pass that wrote Bt
       TAIL_DIRECT_EPOCH = 2, // epoch of the classifier run that left panels to the direct kernel
       TAIL_BAND = 3,         // column span of the middle panel (band width of the matrix)
       TAIL_MFMA_EPOCH = 4,   // epoch of the classifier run that gave panels to the matrix-core kernel
       TAIL_MERGE_EPOCH = 6,  // epoch of the call whose direct panels go to the row-merging kernel (rows share column patterns)
       TAIL_MFMAD_EPOCH = 5,  // ... and some of them fall back to the DIRECT kernel when B holds a non-finite value
       TAIL_ROWS_EPOCH = 7,   // epoch of the call whose direct panels go to the four-rows-per-wave kernel (128+ staged columns)
       TAIL_HDR = 16 };
constexpr int TAIL_PARTS = 1024; // (min, max) column pairs of the column-range pass of a row block (behind the header)
enum { PANEL_DIRECT = 0, PANEL_WINDOW = 1, PANEL_MFMA_W = 2, PANEL_MFMA_D = 3,
       PANEL_CLASS_MASK = 0xff, PANEL_SHARED_ROWS = 0x100 /* flag: the panel's leading rows list the same columns */,
       PANEL_PHASE_SHIFT = 9 /* two bits: (row index where a group of three such rows starts) mod 3 */,
       PANEL_WAVE_ROWS = 0x800 /* flag: columns in runs, or row lengths far apart: a row per wave suits the panel */ };
constexpr int MFMA_MAX_WAVES = 8; // 16 rows per wave: panels of up to 128 rows (taller panels never take the MFMA kernel)
size_t workspace_tail_bytes(int64_t rows);
unsigned long long *panel_stats_device();
hipError_t launch_spmm_mfma(hipStream_t s, int rows, int cols, const int *rowptr, const int *colidx, const double *val,
                            const double *Bt, int64_t ldbt, int n, double alpha, double beta, double *C, int64_t ldc,
                            const int2 *info, const int *tail, const int *cls, int panel_rows, int npanels, int epoch,
                            unsigned long long *stats);

// Experiment / test switches, read from the environment once (kernels.hip); options_reload() re-reads them.
struct Options {
    int spmm_variant = SPMM_VARIANT_AUTO;
    char spmv_variant[16] = {0};          // "" = auto
    bool tier16 = true, tier32 = true;    // staged widths 16 / 32 for n <= 16 / 32 (SBLAS_SPMM_MIN_LDBT: 0 = both, 64 = neither: the 64-column copy)
    unsigned long long max_bt_bytes = 0xffffffffull; // SBLAS_SPMM_MAX_BT_BYTES (tests of the column-chunk loop)
    int direct_lds = -1;                  // SBLAS_DIRECT_LDS
    int direct_map = -1;                  // SBLAS_DIRECT_MAP: 1 interleave, 0 contiguous, -1 by span
    int stage_range = -1;                 // SBLAS_STAGE_RANGE: stage only the rows of B a row block refers to (-1: when the saved staging traffic outweighs the column-range pass)
    int direct_merge = 1;                 // SBLAS_DIRECT_MERGE: 128-column direct panels through the row-merging kernel
    double rows8_min_avg = 256.0;         // SBLAS_ROWS8_MIN_AVG
    float window_density = 0.42f;         // SBLAS_WINDOW_DENSITY: the LDS-tiled kernel's bar, in nonzeros per spanned column of a 16-row slice of a panel
    int panel_rows = 0, panel_groups = 0; // SBLAS_SPMM_PANEL_ROWS
    int tune[4] = {0, 0, 0, 0};           // SBLAS_TUNE
    bool validate = false;                // SBLAS_VALIDATE=1: every SpMM / SpMV call checks the CSR contents first (synchronises; debugging)
    float mfma_min_fill = -1.0f;          // SBLAS_MFMA_MIN_FILL: block fill from which a panel takes the MFMA kernel (< 0: built-in rule)
};
const Options &options();
void options_reload();
void raise_dynamic_lds(const void *fn, size_t bytes);

// A per-matrix plan (capi.hip: sblas_hip_spmm_plan_*): the panel verdicts of one (A, staged width) pair kept in a device
// buffer of the plan's own (same layout as the workspace tail), and what the host learned from them once -- which
// stage-2 kernels have panels at all, the matrix-wide votes, a row block's column range.
struct PlanView {
    int *tail = nullptr;        // TAIL_HDR ints + TAIL_PARTS int2 + one int2 + one int per panel (device)
    int epoch = 0;              // the classifier epoch the verdicts carry
    int info_rows = 0, groups = 0;
    long n_window = 0, n_direct = 0, n_mfma_w = 0, n_mfma_d = 0; // panels per class after the votes
    bool merge = false;         // the call's direct panels go to the row-merging kernel
    bool four_rows = false;     // ... to the four-rows-per-wave kernel (128+ staged columns)
    bool use_range = false;     // stage only the column range [lo, hi] of B
    int nparts = 0;
};
size_t plan_tail_bytes(int64_t rows);
bool classify_worthwhile(int64_t rows, int64_t nnz, int64_t ldbt);
hipError_t plan_build(hipStream_t s, int rows, int cols, int64_t nnz, const int *rowptr, const int *colidx, int64_t ldbt,
                      int variant, bool use_range, PlanView *pv);
hipError_t launch_stage_planned(hipStream_t s, int64_t cols, int64_t n, const double *B, int64_t ldb, double *Bt,
                                int64_t ldbt, const PlanView &pv);
hipError_t launch_stage_range(hipStream_t s, int64_t cols, int64_t n, const double *B, int64_t ldb, double *Bt,
                              int64_t ldbt, int rows, int64_t nnz, const int *rowptr, const int *colidx, int variant,
                              int classify, int *epoch_out);
hipError_t launch_dense_to_rowmajor(hipStream_t s, int64_t cols, int64_t n, const double *B, int64_t ldb,
                                    double *Bt, int64_t ldbt);
hipError_t launch_spmm_rowpanel(hipStream_t s, int rows, int cols, int64_t nnz, const int *rowptr, const int *colidx,
                                const double *val, const double *Bt, int64_t ldbt, int n, double alpha,
                                double beta, double *C, int64_t ldc, int variant, int pre_epoch = 0,
                                const PlanView *pv = nullptr);
hipError_t launch_stage_classify(hipStream_t s, int64_t cols, int64_t n, const double *B, int64_t ldb, double *Bt,
                                 int64_t ldbt, int rows, const int *rowptr, const int *colidx, int variant,
                                 int *epoch_out);
hipError_t launch_scale(hipStream_t s, int64_t rows, int64_t n, double beta, double *C, int64_t ldc);
hipError_t validate_csr(hipStream_t s, int64_t rows, int64_t cols, int64_t nnz, const int *rowptr, const int *colidx, int *bad);
hipError_t panel_stats(unsigned long long out[4], bool reset);
hipError_t launch_spmv(hipStream_t s, int rows, int cols, int64_t nnz, const int *rowptr, const int *colidx,
                       const double *val, const double *x, double alpha, double beta, double *y);
hipError_t launch_axpby(hipStream_t s, int64_t n, double alpha, const double *x, double beta, double *y);
void kernel_events_enable(bool on);
hipError_t kernel_events_last_ms(float *ms);
hipError_t launch_merge_rowblocks(hipStream_t s, int64_t M, int64_t N, int g, const double *const *src,
                                  const int64_t *start, const int64_t *nrows, double alpha, double beta, double *C,
                                  int64_t ldc);

// typed_kernels.hip: the value / index types besides <int32, fp64> (reference utility.h:302-316)
enum { VT_F64 = 0, VT_F32 = 1 };
enum { IT_I32 = 0, IT_I64 = 1 };
int64_t typed_spmm_ldbt(int64_t n);
size_t typed_spmm_workspace(int vt, int64_t cols, int64_t n);
hipError_t launch_typed_spmm(hipStream_t s, int vt, int it, int64_t rows, int64_t cols, int64_t nnz, const void *rowptr,
                             const void *colidx, const void *val, const void *B, int64_t ldb, int64_t n, double alpha,
                             double beta, void *C, int64_t ldc, void *ws);
hipError_t launch_typed_spmv(hipStream_t s, int vt, int it, int64_t rows, const void *rowptr, const void *colidx,
                             const void *val, const void *x, double alpha, double beta, void *y);
hipError_t launch_typed_axpby(hipStream_t s, int vt, int64_t n, double alpha, const void *x, double beta, void *y);
hipError_t launch_typed_sum_replicas(hipStream_t s, int vt, void *const *bufs, int g, int64_t n);
hipError_t launch_typed_merge_rowblocks(hipStream_t s, int vt, int64_t M, int64_t N, int g, const void *const *src,
                                        const int64_t *start_row, const int64_t *num_rows, double alpha, double beta,
                                        void *C, int64_t ldc);

} // namespace sblas
