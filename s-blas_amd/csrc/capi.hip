// capi.hip -- the extern "C" surface declared in include/sblas_hip.h (compute entry points).
// Argument checking happens here, on the host, before any kernel is launched: a bad shape, leading dimension, pointer or
// workspace comes back as SBLAS_E_INVALID / SBLAS_E_WORKSPACE.  The CONTENTS of the index arrays are the caller's
// contract, as with the vendor libraries this replaces (a column index outside [0, cols) is an out-of-bounds read);
// SBLAS_VALIDATE=1 / sblas_hip_debug_validate_csr_i32 check them on the device first, at the price of a synchronisation.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/sblas_hip.h"
#include "kernels.h"

namespace {

struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int dev)
    {
        if (dev < 0) return;
        err = hipGetDevice(&prev);
        if (err != hipSuccess) return;
        if (prev != dev) {
            err = hipSetDevice(dev);
            switched = (err == hipSuccess);
        }
    }
    ~DeviceScope()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

inline int spmm_variant() { return sblas::options().spmm_variant; }
// the device a `dev` argument means (dev < 0: the calling thread's current device); -1 when that cannot be told
inline int resolve_device(int dev)
{
    if (dev >= 0) return dev;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return -1;
    return cur;
}

inline bool csr_args_ok(int64_t rows, int64_t cols, int64_t nnz, const void *rowptr, const void *colidx,
                        const void *val)
{
    if (rows < 0 || cols < 0 || nnz < 0) return false;
    if (rows > INT_MAX - 64 || cols > INT_MAX || nnz > INT_MAX) return false; // int32 index API
    if (!rowptr) return false;
    if (nnz > 0 && (!colidx || !val)) return false;
    return true;
}

} // namespace

extern "C" {

int sblas_hip_version(void) { return 100; }

const char *sblas_hip_error_string(int code)
{
    switch (code) {
    case SBLAS_OK: return "success";
    case SBLAS_E_INVALID: return "invalid argument";
    case SBLAS_E_HIP: return "HIP runtime / kernel launch failure";
    case SBLAS_E_WORKSPACE: return "workspace missing or too small";
    case SBLAS_E_RCCL: return "RCCL unavailable or collective failed";
    case SBLAS_E_IO: return "MatrixMarket read/parse failure";
    case SBLAS_E_NOGPU: return "no HIP device";
    default: return "unknown sblas error";
    }
}

int sblas_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// staged width of an n-column block: 8, 16 (tier16), 32 (tier32), 64 or a multiple of 128.  The 16- / 32-column tiers are
// also the last resort when even a 64-column staging copy would exceed the 32-bit offset window (chunk_ldbt).
static int64_t ldbt_pick(int64_t n, bool tier16, bool tier32)
{
    if (n <= 0) return 0;
    if (n <= 8) return 8;
    if (n <= 16 && tier16) return 16;
    if (n <= 32 && tier32) return 32;
    if (n <= 64) return 64;
    return (n + 127) / 128 * 128; // wide tiles are 128 columns (two per lane)
}

int64_t sblas_hip_spmm_ldbt(int64_t n)
{
    // 8 / 16 / 32 / 64 staged columns for n <= 8 / 16 / 32 / 64: method 1 on 8 / 4 / 2 GPUs hands a GPU exactly these
    // widths of a 64-column B (matrix.h:554-568), and the narrow LDS-tiled kernel (spmm_lanes_kernel) reads NC * 8 bytes
    // of LDS per nonzero instead of the 512 of a zero-padded 64-column tile (banded bench matrix, N = 8 / 16 / 32 on the
    // 64-column path: 0.265 / 0.238 / 0.245 ms).  SBLAS_SPMM_MIN_LDBT=64 puts 9..32 columns back on the 64-column
    // staging copy (A/B runs, tests).
    return ldbt_pick(n, sblas::options().tier16, sblas::options().tier32);
}
static bool ldbt_ok(int64_t ldbt, int64_t n)
{
    return ldbt == ldbt_pick(n, false, false) || ldbt == ldbt_pick(n, true, true) || ldbt == ldbt_pick(n, true, false) ||
           ldbt == ldbt_pick(n, false, true);
}

// The kernels address Bt with 32-bit byte offsets, so one stage-2 launch can cover at most 4 GiB of Bt.  The
// top-level call therefore walks the dense columns in chunks of `w` columns with (cols+1)*ldbt(w)*8 <= 4 GiB
// (Queen_4147 with N = 256: two chunks of 128).  SBLAS_SPMM_MAX_BT_BYTES lowers the limit (tests).
static uint64_t bt_byte_limit() { return sblas::options().max_bt_bytes; }

// Staging traffic saved if a row block's columns span no more than twice its rows, against the extra pass over the column
// indices and one more launch (~5 us = 40 MB at staging speed; a quarter of nd24k at N = 128 breaks even, an eighth of a
// Queen-like matrix at N = 256 runs 1.5x faster: tools/spmm_shapes.py --block)
static bool range_staging_pays(int64_t rows, int64_t cols, int64_t nnz, int64_t ldbt)
{
    return cols > 2 * rows && (uint64_t)(cols - 2 * rows) * (uint64_t)ldbt * 16ull > (uint64_t)nnz * 8ull + (40ull << 20);
}

static int64_t spmm_chunk_cols(int64_t cols, int64_t n)
{
    const uint64_t lim = bt_byte_limit();
    const uint64_t row_bytes = 8ull * ((uint64_t)cols + 1);
    if (row_bytes * (uint64_t)sblas_hip_spmm_ldbt(n) <= lim) return n;
    for (int64_t w = (n / 128) * 128; w >= 128; w -= 128)
        if (row_bytes * (uint64_t)w <= lim) return w;
    if (row_bytes * 64ull <= lim) return 64;
    return 32; // the narrow kernels use 64-bit addressing: no limit
}
// ldbt of a chunk of nj columns: the narrow tier when the chunk width itself was forced down to 32
static int64_t chunk_ldbt(int64_t cols, int64_t n, int64_t nj)
{
    const bool forced_narrow = spmm_chunk_cols(cols, n) <= 32 && 8ull * ((uint64_t)cols + 1) * 64ull > bt_byte_limit();
    return forced_narrow ? ldbt_pick(nj, true, true) : sblas_hip_spmm_ldbt(nj);
}

size_t sblas_hip_spmm_csr_f64_i32_workspace(int64_t rows, int64_t cols, int64_t nnz, int64_t n)
{
    (void)nnz;
    if (cols <= 0 || n <= 0) return 0;
    // Bt for one column chunk plus one extra all-zero row (the target of masked DPP slots), then one int2 per
    // row panel (the panel classifier's verdicts)
    const int64_t w = spmm_chunk_cols(cols, n);
    const size_t bt = ((size_t)cols + 1) * (size_t)chunk_ldbt(cols, n, w) * sizeof(double);
    return bt + sblas::workspace_tail_bytes(rows); // flags, one span and one class per row panel
}

int sblas_hip_dense_to_rowmajor_f64(int dev, void *stream, int64_t cols, int64_t n, const double *B,
                                    int64_t ldb, double *Bt, int64_t ldbt)
{
    if (cols < 0 || n < 0) return SBLAS_E_INVALID;
    if (cols == 0 || n == 0) return SBLAS_OK;
    if (!B || !Bt || ldb < cols || ldbt < n || !ldbt_ok(ldbt, n)) return SBLAS_E_INVALID;
    DeviceScope scope(dev);
    if (scope.err != hipSuccess) return SBLAS_E_HIP;
    return sblas::launch_dense_to_rowmajor((hipStream_t)stream, cols, n, B, ldb, Bt, ldbt) == hipSuccess
               ? SBLAS_OK
               : SBLAS_E_HIP;
}

// A has no nonzeros (cols == 0 or nnz == 0): C = beta * C, nothing else.  Neither the staging copy nor the panel
// verdicts behind it are touched, so a NULL / empty workspace is fine here.
static int scale_only(int dev, void *stream, int64_t rows, int64_t n, double beta, double *C, int64_t ldc)
{
    if (beta == 1.0) return SBLAS_OK;
    DeviceScope scope(dev);
    if (scope.err != hipSuccess) return SBLAS_E_HIP;
    return sblas::launch_scale((hipStream_t)stream, rows, n, beta, C, ldc) == hipSuccess ? SBLAS_OK : SBLAS_E_HIP;
}

static int spmm_staged(int dev, void *stream, int64_t rows, int64_t cols, int64_t nnz, const int32_t *rowptr,
                       const int32_t *colidx, const double *val, const double *Bt, int64_t ldbt, int64_t n,
                       double alpha, double beta, double *C, int64_t ldc, int pre_epoch)
{
    if (!csr_args_ok(rows, cols, nnz, rowptr, colidx, val) || n < 0) return SBLAS_E_INVALID;
    if (rows == 0 || n == 0) return SBLAS_OK;
    if (!C || ldc < rows || n > INT_MAX) return SBLAS_E_INVALID;
    if (!ldbt_ok(ldbt, n)) return SBLAS_E_INVALID;
    if (cols == 0 || nnz == 0) return scale_only(dev, stream, rows, n, beta, C, ldc); // A*B = 0: no kernel reads Bt
    if (!Bt) return SBLAS_E_INVALID;
    // the kernels address Bt with 32-bit element offsets (row * ldbt + column)
    if (ldbt >= 64 && ((uint64_t)cols + 1) * (uint64_t)ldbt * 8ull > 0xffffffffull) return SBLAS_E_INVALID; // 32-bit byte offsets
    DeviceScope scope(dev);
    if (scope.err != hipSuccess) return SBLAS_E_HIP;
    if ((reinterpret_cast<uintptr_t>(Bt) & 15u) != 0) return SBLAS_E_INVALID; // 16-byte tile loads
    return sblas::launch_spmm_rowpanel((hipStream_t)stream, (int)rows, (int)cols, nnz, rowptr, colidx, val, Bt, ldbt,
                                       (int)n, alpha, beta, C, ldc, spmm_variant(), pre_epoch) == hipSuccess
               ? SBLAS_OK
               : SBLAS_E_HIP;
}

int sblas_hip_spmm_csr_rowmajorB_f64_i32(int dev, void *stream, int64_t rows, int64_t cols, int64_t nnz,
                                         const int32_t *rowptr, const int32_t *colidx, const double *val,
                                         const double *Bt, int64_t ldbt, int64_t n, double alpha,
                                         double beta, double *C, int64_t ldc)
{
    return spmm_staged(dev, stream, rows, cols, nnz, rowptr, colidx, val, Bt, ldbt, n, alpha, beta, C, ldc, 0);
}

static int validate_if_asked(int dev, void *stream, int64_t rows, int64_t cols, int64_t nnz, const int32_t *rowptr,
                             const int32_t *colidx);

// A plan: the panel verdicts of one matrix structure at one staged width, taken once (sblas_hip_spmm_plan_create).
struct SpmmPlan {
    int dev = -1;
    int64_t rows = 0, cols = 0, nnz = 0, n = 0, ldbt = 0;
    const void *rowptr = nullptr, *colidx = nullptr;
    bool active = false;      // false: nothing to plan (empty matrix, a pinned direct variant): calls run unplanned
    void *buf = nullptr;
    sblas::PlanView pv;
};

static int spmm_impl(int dev, void *stream, int64_t rows, int64_t cols, int64_t nnz, const int32_t *rowptr,
                     const int32_t *colidx, const double *val, const double *B, int64_t ldb, int64_t n, double alpha,
                     double beta, double *C, int64_t ldc, void *workspace, size_t workspace_bytes, const SpmmPlan *plan)
{
    if (!csr_args_ok(rows, cols, nnz, rowptr, colidx, val) || n < 0) return SBLAS_E_INVALID;
    if (rows == 0 || n == 0) return SBLAS_OK;
    if (!C || ldc < rows) return SBLAS_E_INVALID;
    if (cols > 0 && (!B || ldb < cols)) return SBLAS_E_INVALID;
    if (cols == 0 || nnz == 0) return scale_only(dev, stream, rows, n, beta, C, ldc);
    if (const int vrc = validate_if_asked(dev, stream, rows, cols, nnz, rowptr, colidx)) return vrc;
    const size_t need = sblas_hip_spmm_csr_f64_i32_workspace(rows, cols, nnz, n);
    if (need > 0 && (!workspace || workspace_bytes < need)) return SBLAS_E_WORKSPACE;
    double *Bt = static_cast<double *>(workspace);
    const int64_t w = spmm_chunk_cols(cols, n);
    int range_epoch = 0;      // range staging: the first chunk's column range and panel verdicts serve the later chunks
    int64_t range_ldbt = 0;
    for (int64_t j0 = 0; j0 < n; j0 += w) { // one pass unless Bt would exceed the 32-bit offset window
        const int64_t nj = (n - j0 < w) ? n - j0 : w;
        const int64_t ldbt = chunk_ldbt(cols, n, nj);
        int rc, pre_epoch = 0;
        const int sr = sblas::options().stage_range;
        // Staging traffic saved if the block's columns span no more than twice its rows, against the extra pass over the
        // column indices and one more launch (~5 us = 40 MB at staging speed; a quarter of nd24k at N = 128 breaks even,
        // an eighth of a Queen-like matrix at N = 256 runs 1.5x faster: tools/spmm_shapes.py --block)
        const bool pays = range_staging_pays(rows, cols, nnz, ldbt);
        if (plan && plan->active && ldbt == plan->ldbt && ldbt_ok(ldbt, nj)) {
            // planned: stage (no classifier rides along), then only the stage-2 kernels that have panels
            DeviceScope scope(dev);
            if (scope.err != hipSuccess) return SBLAS_E_HIP;
            if ((reinterpret_cast<uintptr_t>(Bt) & 15u) != 0) return SBLAS_E_INVALID;
            if (sblas::launch_stage_planned((hipStream_t)stream, cols, nj, B + j0 * ldb, ldb, Bt, ldbt, plan->pv) != hipSuccess)
                return SBLAS_E_HIP;
            if (sblas::launch_spmm_rowpanel((hipStream_t)stream, (int)rows, (int)cols, nnz, rowptr, colidx, val, Bt, ldbt, (int)nj,
                                            alpha, beta, C + j0 * ldc, ldc, spmm_variant(), 0, &plan->pv) != hipSuccess)
                return SBLAS_E_HIP;
            continue;
        }
        if (ldbt >= 64 && ldbt_ok(ldbt, nj) && (sr > 0 || (sr < 0 && pays))) {
            // a row block (method 2): the row-major copy covers only the rows of B the block's nonzeros refer to
            DeviceScope scope(dev);
            if (scope.err != hipSuccess) return SBLAS_E_HIP;
            const bool direct_only = spmm_variant() == sblas::SPMM_VARIANT_DIRECT_DPP ||
                                     spmm_variant() == sblas::SPMM_VARIANT_DIRECT_ROWS ||
                                     spmm_variant() == sblas::SPMM_VARIANT_DIRECT_MERGE ||
                                     !sblas::classify_worthwhile(rows, nnz, ldbt);
            pre_epoch = (!direct_only && ldbt == range_ldbt) ? range_epoch : 0;
            if (sblas::launch_stage_range((hipStream_t)stream, cols, nj, B + j0 * ldb, ldb, Bt, ldbt, (int)rows, nnz,
                                          rowptr, colidx, spmm_variant(), direct_only ? 0 : 1, &pre_epoch) != hipSuccess)
                return SBLAS_E_HIP;
            range_epoch = pre_epoch;
            range_ldbt = ldbt;
        } else if (spmm_variant() != sblas::SPMM_VARIANT_DIRECT_DPP && spmm_variant() != sblas::SPMM_VARIANT_DIRECT_ROWS &&
            spmm_variant() != sblas::SPMM_VARIANT_LANES &&
            sblas::classify_worthwhile(rows, nnz, ldbt) &&
            (ldbt >= 64 || (spmm_variant() != sblas::SPMM_VARIANT_DIRECT_MERGE &&
                            ((uint64_t)cols + 1) * (uint64_t)ldbt * 8ull <= 0xffffffffull)) &&
            cols > 0 && nnz > 0 && ldb >= cols && ldbt_ok(ldbt, nj)) {
            // default path: the panel classifier rides in the staging launch (one launch and one gap less per call)
            DeviceScope scope(dev);
            if (scope.err != hipSuccess) return SBLAS_E_HIP;
            if (sblas::launch_stage_classify((hipStream_t)stream, cols, nj, B + j0 * ldb, ldb, Bt, ldbt, (int)rows, rowptr,
                                             colidx, spmm_variant(), &pre_epoch) != hipSuccess)
                return SBLAS_E_HIP;
        } else {
            rc = sblas_hip_dense_to_rowmajor_f64(dev, stream, cols, nj, B + j0 * ldb, ldb, Bt, ldbt);
            if (rc != SBLAS_OK) return rc;
        }
        rc = spmm_staged(dev, stream, rows, cols, nnz, rowptr, colidx, val, Bt, ldbt, nj, alpha, beta, C + j0 * ldc, ldc,
                         pre_epoch);
        if (rc != SBLAS_OK) return rc;
    }
    return SBLAS_OK;
}

int sblas_hip_spmm_csr_f64_i32(int dev, void *stream, int64_t rows, int64_t cols, int64_t nnz,
                               const int32_t *rowptr, const int32_t *colidx, const double *val,
                               const double *B, int64_t ldb, int64_t n, double alpha, double beta,
                               double *C, int64_t ldc, void *workspace, size_t workspace_bytes)
{
    return spmm_impl(dev, stream, rows, cols, nnz, rowptr, colidx, val, B, ldb, n, alpha, beta, C, ldc, workspace,
                     workspace_bytes, nullptr);
}

// ---- per-matrix plan (the slot of cusparseSpMM_bufferSize / preprocess, spmm.h:134-141) --------------------------
int sblas_hip_spmm_plan_create(int dev, void *stream, int64_t rows, int64_t cols, int64_t nnz, const int32_t *rowptr,
                               const int32_t *colidx, int64_t n, void **plan_out)
{
    if (!plan_out || !csr_args_ok(rows, cols, nnz, rowptr, colidx, reinterpret_cast<const void *>(1)) || n < 0) return SBLAS_E_INVALID;
    SpmmPlan *p = new SpmmPlan;
    p->dev = resolve_device(dev), p->rows = rows, p->cols = cols, p->nnz = nnz, p->n = n, p->rowptr = rowptr, p->colidx = colidx;
    *plan_out = p;
    const int v = spmm_variant();
    const bool classified = v == sblas::SPMM_VARIANT_AUTO || v == sblas::SPMM_VARIANT_MFMA || v == sblas::SPMM_VARIANT_NO_MFMA;
    if (rows == 0 || cols == 0 || nnz == 0 || n == 0 || !classified) return SBLAS_OK; // nothing to plan
    const int64_t w = spmm_chunk_cols(cols, n);
    const int64_t ldbt = chunk_ldbt(cols, n, n < w ? n : w);
    if (ldbt < 64 && ((uint64_t)cols + 1) * (uint64_t)ldbt * 8ull > 0xffffffffull) return SBLAS_OK; // 64-bit narrow kernels: unplanned
    if (!sblas::classify_worthwhile(rows, nnz, ldbt)) return SBLAS_OK; // short rows at a width of 64 columns or fewer: nothing is classified
    DeviceScope scope(dev);
    if (scope.err != hipSuccess) { delete p; *plan_out = nullptr; return SBLAS_E_HIP; }
    if (hipMalloc(&p->buf, sblas::plan_tail_bytes(rows)) != hipSuccess) { delete p; *plan_out = nullptr; return SBLAS_E_HIP; }
    p->pv.tail = static_cast<int *>(p->buf);
    p->ldbt = ldbt;
    const int sr = sblas::options().stage_range;
    const bool use_range = ldbt >= 64 && (sr > 0 || (sr < 0 && range_staging_pays(rows, cols, nnz, ldbt)));
    if (sblas::plan_build((hipStream_t)stream, (int)rows, (int)cols, nnz, rowptr, colidx, ldbt, v, use_range, &p->pv) != hipSuccess) {
        (void)hipFree(p->buf);
        delete p;
        *plan_out = nullptr;
        return SBLAS_E_HIP;
    }
    p->active = true;
    return SBLAS_OK;
}

int sblas_hip_spmm_plan_destroy(void *plan)
{
    if (!plan) return SBLAS_OK;
    SpmmPlan *p = static_cast<SpmmPlan *>(plan);
    if (p->buf) {
        DeviceScope scope(p->dev);
        (void)hipFree(p->buf);
    }
    delete p;
    return SBLAS_OK;
}

int sblas_hip_spmm_plan_info(const void *plan, int64_t out[8])
{
    if (!plan || !out) return SBLAS_E_INVALID;
    const SpmmPlan *p = static_cast<const SpmmPlan *>(plan);
    out[0] = p->active, out[1] = p->pv.n_window, out[2] = p->pv.n_direct, out[3] = p->pv.n_mfma_w + p->pv.n_mfma_d;
    out[4] = p->pv.merge ? 1 : p->pv.four_rows ? 2 : 0, out[5] = p->pv.use_range, out[6] = p->ldbt, out[7] = p->pv.info_rows;
    return SBLAS_OK;
}

int sblas_hip_spmm_csr_f64_i32_planned(const void *plan, int dev, void *stream, int64_t rows, int64_t cols, int64_t nnz,
                                       const int32_t *rowptr, const int32_t *colidx, const double *val, const double *B,
                                       int64_t ldb, int64_t n, double alpha, double beta, double *C, int64_t ldc,
                                       void *workspace, size_t workspace_bytes)
{
    if (!plan) return SBLAS_E_INVALID;
    const SpmmPlan *p = static_cast<const SpmmPlan *>(plan);
    // the plan speaks for ONE structure: the same arrays it was made from (their contents are the caller's promise)
    // ... on the device its verdicts live on
    if (p->dev != resolve_device(dev) || p->rows != rows || p->cols != cols || p->nnz != nnz || p->rowptr != rowptr || p->colidx != colidx)
        return SBLAS_E_INVALID;
    return spmm_impl(dev, stream, rows, cols, nnz, rowptr, colidx, val, B, ldb, n, alpha, beta, C, ldc, workspace, workspace_bytes, p);
}

int sblas_hip_debug_validate_csr_i32(int dev, void *stream, int64_t rows, int64_t cols, int64_t nnz, const int32_t *rowptr,
                                     const int32_t *colidx)
{
    if (!csr_args_ok(rows, cols, nnz, rowptr, colidx, reinterpret_cast<const void *>(1))) return SBLAS_E_INVALID;
    if (rows == 0) return SBLAS_OK;
    DeviceScope scope(dev);
    if (scope.err != hipSuccess) return SBLAS_E_HIP;
    int bad = 0;
    if (sblas::validate_csr((hipStream_t)stream, rows, cols, nnz, rowptr, colidx, &bad) != hipSuccess) return SBLAS_E_HIP;
    return bad ? SBLAS_E_INVALID : SBLAS_OK;
}
static int validate_if_asked(int dev, void *stream, int64_t rows, int64_t cols, int64_t nnz, const int32_t *rowptr,
                             const int32_t *colidx)
{
    if (!sblas::options().validate || rows <= 0 || nnz <= 0) return SBLAS_OK;
    return sblas_hip_debug_validate_csr_i32(dev, stream, rows, cols, nnz, rowptr, colidx);
}

int sblas_hip_debug_reload_env(void)
{
    sblas::options_reload();
    return SBLAS_OK;
}

int sblas_hip_debug_spmm_kernel_events(int enable)
{
    sblas::kernel_events_enable(enable != 0);
    return SBLAS_OK;
}

int sblas_hip_debug_spmm_last_kernel_ms(float *ms)
{
    if (!ms) return SBLAS_E_INVALID;
    return sblas::kernel_events_last_ms(ms) == hipSuccess ? SBLAS_OK : SBLAS_E_HIP;
}

int sblas_hip_debug_spmm_panel_stats(uint64_t out[4], int reset)
{
    if (!out) return SBLAS_E_INVALID;
    unsigned long long tmp[4];
    if (sblas::panel_stats(tmp, reset != 0) != hipSuccess) return SBLAS_E_HIP;
    for (int i = 0; i < 4; ++i) out[i] = tmp[i];
    return SBLAS_OK;
}

int sblas_hip_spmv_csr_f64_i32(int dev, void *stream, int64_t rows, int64_t cols, int64_t nnz,
                               const int32_t *rowptr, const int32_t *colidx, const double *val,
                               const double *x, double alpha, double beta, double *y)
{
    if (!csr_args_ok(rows, cols, nnz, rowptr, colidx, val)) return SBLAS_E_INVALID;
    if (rows == 0) return SBLAS_OK;
    if (!y || (cols > 0 && !x)) return SBLAS_E_INVALID;
    if (const int vrc = validate_if_asked(dev, stream, rows, cols, nnz, rowptr, colidx)) return vrc;
    DeviceScope scope(dev);
    if (scope.err != hipSuccess) return SBLAS_E_HIP;
    return sblas::launch_spmv((hipStream_t)stream, (int)rows, (int)cols, nnz, rowptr, colidx, val, x, alpha, beta, y) ==
                   hipSuccess
               ? SBLAS_OK
               : SBLAS_E_HIP;
}

int sblas_hip_axpby_f64(int dev, void *stream, int64_t n, double alpha, const double *x, double beta,
                        double *y)
{
    if (n < 0) return SBLAS_E_INVALID;
    if (n == 0) return SBLAS_OK;
    if (!x || !y) return SBLAS_E_INVALID;
    DeviceScope scope(dev);
    if (scope.err != hipSuccess) return SBLAS_E_HIP;
    return sblas::launch_axpby((hipStream_t)stream, n, alpha, x, beta, y) == hipSuccess ? SBLAS_OK
                                                                                       : SBLAS_E_HIP;
}

// ---- the other value / index types (typed_kernels.hip); <int32, fp64> forwards to the tuned entry points above ----
static bool types_ok(int vtype, int itype)
{
    return (vtype == SBLAS_F64 || vtype == SBLAS_F32) && (itype == SBLAS_I32 || itype == SBLAS_I64);
}

size_t sblas_hip_spmm_csr_workspace(int vtype, int itype, int64_t rows, int64_t cols, int64_t nnz, int64_t n)
{
    if (!types_ok(vtype, itype)) return 0;
    if (vtype == SBLAS_F64 && itype == SBLAS_I32) return sblas_hip_spmm_csr_f64_i32_workspace(rows, cols, nnz, n);
    return sblas::typed_spmm_workspace(vtype, cols, n);
}

int sblas_hip_spmm_csr(int dev, void *stream, int vtype, int itype, int64_t rows, int64_t cols, int64_t nnz,
                       const void *rowptr, const void *colidx, const void *val, const void *B, int64_t ldb, int64_t n,
                       double alpha, double beta, void *C, int64_t ldc, void *workspace, size_t workspace_bytes)
{
    if (!types_ok(vtype, itype)) return SBLAS_E_INVALID;
    if (vtype == SBLAS_F64 && itype == SBLAS_I32)
        return sblas_hip_spmm_csr_f64_i32(dev, stream, rows, cols, nnz, static_cast<const int32_t *>(rowptr),
                                          static_cast<const int32_t *>(colidx), static_cast<const double *>(val),
                                          static_cast<const double *>(B), ldb, n, alpha, beta, static_cast<double *>(C),
                                          ldc, workspace, workspace_bytes);
    if (rows < 0 || cols < 0 || nnz < 0 || n < 0 || !rowptr || (nnz > 0 && (!colidx || !val))) return SBLAS_E_INVALID;
    if (itype == SBLAS_I32 && (rows > INT_MAX - 64 || cols > INT_MAX || nnz > INT_MAX)) return SBLAS_E_INVALID;
    if (rows == 0 || n == 0) return SBLAS_OK;
    if (!C || ldc < rows) return SBLAS_E_INVALID;
    if (cols > 0 && (!B || ldb < cols)) return SBLAS_E_INVALID;
    const size_t need = (cols == 0 || nnz == 0) ? 0 : sblas::typed_spmm_workspace(vtype, cols, n);
    if (need > 0 && (!workspace || workspace_bytes < need)) return SBLAS_E_WORKSPACE;
    DeviceScope scope(dev);
    if (scope.err != hipSuccess) return SBLAS_E_HIP;
    return sblas::launch_typed_spmm((hipStream_t)stream, vtype, itype, rows, cols, nnz, rowptr, colidx, val, B, ldb, n, alpha,
                                    beta, C, ldc, workspace) == hipSuccess
               ? SBLAS_OK
               : SBLAS_E_HIP;
}

int sblas_hip_spmv_csr(int dev, void *stream, int vtype, int itype, int64_t rows, int64_t cols, int64_t nnz,
                       const void *rowptr, const void *colidx, const void *val, const void *x, double alpha,
                       double beta, void *y)
{
    if (!types_ok(vtype, itype)) return SBLAS_E_INVALID;
    if (vtype == SBLAS_F64 && itype == SBLAS_I32)
        return sblas_hip_spmv_csr_f64_i32(dev, stream, rows, cols, nnz, static_cast<const int32_t *>(rowptr),
                                          static_cast<const int32_t *>(colidx), static_cast<const double *>(val),
                                          static_cast<const double *>(x), alpha, beta, static_cast<double *>(y));
    if (rows < 0 || cols < 0 || nnz < 0 || !rowptr || (nnz > 0 && (!colidx || !val))) return SBLAS_E_INVALID;
    if (itype == SBLAS_I32 && (rows > INT_MAX - 64 || cols > INT_MAX || nnz > INT_MAX)) return SBLAS_E_INVALID;
    if (rows == 0) return SBLAS_OK;
    if (!y || (cols > 0 && !x)) return SBLAS_E_INVALID;
    DeviceScope scope(dev);
    if (scope.err != hipSuccess) return SBLAS_E_HIP;
    return sblas::launch_typed_spmv((hipStream_t)stream, vtype, itype, rows, rowptr, colidx, val, x, alpha, beta, y) ==
                   hipSuccess
               ? SBLAS_OK
               : SBLAS_E_HIP;
}

int sblas_hip_axpby(int dev, void *stream, int vtype, int64_t n, double alpha, const void *x, double beta, void *y)
{
    if (vtype == SBLAS_F64)
        return sblas_hip_axpby_f64(dev, stream, n, alpha, static_cast<const double *>(x), beta, static_cast<double *>(y));
    if (vtype != SBLAS_F32 || n < 0) return SBLAS_E_INVALID;
    if (n == 0) return SBLAS_OK;
    if (!x || !y) return SBLAS_E_INVALID;
    DeviceScope scope(dev);
    if (scope.err != hipSuccess) return SBLAS_E_HIP;
    return sblas::launch_typed_axpby((hipStream_t)stream, vtype, n, alpha, x, beta, y) == hipSuccess ? SBLAS_OK : SBLAS_E_HIP;
}

} // extern "C"
