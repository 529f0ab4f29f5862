// comparison only (SURVEY 8d "optional vendor cross-check"): rocSPARSE's generic CSR SpMM / SpMV on the bench matrix,
// same layouts as the product (B, C column-major fp64, int32 indices), timed with HIP events.  Not linked into the
// product, not used by any test.  Input: the binary CSR dump written by tools/rocsparse_compare.sh.
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { fprintf(stderr, "%s failed: %d (line %d)\n", #x, (int)e_, __LINE__); exit(1); } } while (0)
int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s csr.bin N\n", argv[0]); return 2; }
    const int n = atoi(argv[2]);
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    long long h[3];
    if (fread(h, 8, 3, f) != 3) return 1;
    const long long rows = h[0], cols = h[1], nnz = h[2];
    std::vector<int> rp(rows + 1), ci(nnz);
    std::vector<double> v(nnz);
    if (fread(rp.data(), 4, rows + 1, f) != (size_t)rows + 1 || fread(ci.data(), 4, nnz, f) != (size_t)nnz || fread(v.data(), 8, nnz, f) != (size_t)nnz) return 1;
    fclose(f);
    int *d_rp, *d_ci; double *d_v, *d_B, *d_C, *d_x, *d_y;
    CK(hipMalloc(&d_rp, (rows + 1) * 4)); CK(hipMalloc(&d_ci, nnz * 4)); CK(hipMalloc(&d_v, nnz * 8));
    CK(hipMalloc(&d_B, cols * n * 8)); CK(hipMalloc(&d_C, rows * n * 8)); CK(hipMalloc(&d_x, cols * 8)); CK(hipMalloc(&d_y, rows * 8));
    CK(hipMemcpy(d_rp, rp.data(), (rows + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ci, ci.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_v, v.data(), nnz * 8, hipMemcpyHostToDevice));
    std::vector<double> ones((size_t)std::max(rows, cols) * n, 1.0);
    CK(hipMemcpy(d_B, ones.data(), cols * n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_C, ones.data(), rows * n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_x, ones.data(), cols * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_y, ones.data(), rows * 8, hipMemcpyHostToDevice));
    rocsparse_handle handle; CK(rocsparse_create_handle(&handle));
    rocsparse_spmat_descr A; rocsparse_dnmat_descr B, C; rocsparse_dnvec_descr x, y;
    CK(rocsparse_create_csr_descr(&A, rows, cols, nnz, d_rp, d_ci, d_v, rocsparse_indextype_i32, rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    CK(rocsparse_create_dnmat_descr(&B, cols, n, cols, d_B, rocsparse_datatype_f64_r, rocsparse_order_column));
    CK(rocsparse_create_dnmat_descr(&C, rows, n, rows, d_C, rocsparse_datatype_f64_r, rocsparse_order_column));
    CK(rocsparse_create_dnvec_descr(&x, cols, d_x, rocsparse_datatype_f64_r)); CK(rocsparse_create_dnvec_descr(&y, rows, d_y, rocsparse_datatype_f64_r));
    const double alpha = 1.0, beta = 1.0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { rocsparse_spmm_alg alg; const char *name; } algs[] = {{rocsparse_spmm_alg_default, "default"}, {rocsparse_spmm_alg_csr, "csr"},
        {rocsparse_spmm_alg_csr_row_split, "csr_row_split"}, {rocsparse_spmm_alg_csr_nnz_split, "csr_nnz_split"}, {rocsparse_spmm_alg_csr_merge_path, "csr_merge_path"}};
    for (auto &a : algs) {
        size_t bytes = 0; void *buf = nullptr;
        if (rocsparse_spmm(handle, rocsparse_operation_none, rocsparse_operation_none, &alpha, A, B, &beta, C, rocsparse_datatype_f64_r, a.alg, rocsparse_spmm_stage_buffer_size, &bytes, nullptr) != 0) { printf("rocsparse_spmm %-15s: not supported\n", a.name); continue; }
        CK(hipMalloc(&buf, bytes ? bytes : 8));
        hipEventRecord(e0);
        CK(rocsparse_spmm(handle, rocsparse_operation_none, rocsparse_operation_none, &alpha, A, B, &beta, C, rocsparse_datatype_f64_r, a.alg, rocsparse_spmm_stage_preprocess, &bytes, buf));
        hipEventRecord(e1); hipEventSynchronize(e1); float pre; hipEventElapsedTime(&pre, e0, e1);
        for (int i = 0; i < 100; ++i) CK(rocsparse_spmm(handle, rocsparse_operation_none, rocsparse_operation_none, &alpha, A, B, &beta, C, rocsparse_datatype_f64_r, a.alg, rocsparse_spmm_stage_compute, &bytes, buf));
        const int reps = 20;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) CK(rocsparse_spmm(handle, rocsparse_operation_none, rocsparse_operation_none, &alpha, A, B, &beta, C, rocsparse_datatype_f64_r, a.alg, rocsparse_spmm_stage_compute, &bytes, buf));
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("rocsparse_spmm %-15s N=%d: %.3f ms per call = %.1f GFLOP/s (preprocess %.3f ms, buffer %zu bytes)\n", a.name, n, ms, 2.0 * nnz * n / ms / 1e6, pre, bytes);
        hipFree(buf);
    }
    {
        size_t bytes = 0; void *buf = nullptr;
        CK(rocsparse_spmv(handle, rocsparse_operation_none, &alpha, A, x, &beta, y, rocsparse_datatype_f64_r, rocsparse_spmv_alg_default, rocsparse_spmv_stage_buffer_size, &bytes, nullptr));
        CK(hipMalloc(&buf, bytes ? bytes : 8));
        CK(rocsparse_spmv(handle, rocsparse_operation_none, &alpha, A, x, &beta, y, rocsparse_datatype_f64_r, rocsparse_spmv_alg_default, rocsparse_spmv_stage_preprocess, &bytes, buf));
        for (int i = 0; i < 500; ++i) CK(rocsparse_spmv(handle, rocsparse_operation_none, &alpha, A, x, &beta, y, rocsparse_datatype_f64_r, rocsparse_spmv_alg_default, rocsparse_spmv_stage_compute, &bytes, buf));
        const int reps = 50;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) CK(rocsparse_spmv(handle, rocsparse_operation_none, &alpha, A, x, &beta, y, rocsparse_datatype_f64_r, rocsparse_spmv_alg_default, rocsparse_spmv_stage_compute, &bytes, buf));
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("rocsparse_spmv default: %.1f us per call = %.2f TB/s algorithmic\n", ms * 1e3, (nnz * 12.0 + rows * 24.0 + cols * 8.0) / ms / 1e9);
    }
    return 0;
}
