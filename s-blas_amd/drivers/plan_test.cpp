// plan_test -- the SpMM ops called repeatedly on one matrix, as an iterative caller does: from the second call on the
// header layer makes a per-matrix plan (spmm.h: spmm_on_gpu) and the later calls run planned.  Every call is checked
// against the host verifier.   plan_test <matrix> <B_width> <gpus> [calls]
#include "harness.h"

static bool repeated(int method, const char *path, int b_width, unsigned n_gpu, int calls)
{
    CsrSparseMatrix<int, double> A(path);
    if (A.height == 0 || A.nnz == 0) return false;
    DenseMatrix<int, double> B(A.width, b_width, col_major);
    A.sync2gpu(n_gpu, method == 1 ? replicate : segment);
    B.sync2gpu(n_gpu, method == 1 ? segment : replicate);
    bool ok = true;
    for (int c = 0; c < calls; ++c) {
        const double alpha = 1.0 + c, beta = 0.5 * c;
        DenseMatrix<int, double> C(A.height, b_width, 1.0, col_major), C_cpu(A.height, b_width, 1.0, col_major);
        C.sync2gpu(n_gpu, method == 1 ? segment : replicate);
        if (method == 1) sblas_spmm_csr_v1<int, double>(&A, &B, &C, alpha, beta, n_gpu);
        else sblas_spmm_csr_v2<int, double>(&A, &B, &C, alpha, beta, n_gpu);
        CUDA_CHECK_ERROR();
        if (method == 2) C.sync2cpu(0);
        sblas_spmm_csr_cpu<int, double>(&A, &B, &C_cpu, alpha, beta);
        const harness::Outcome o = harness::compare(C_cpu.val, C.val, C.get_mtx_num());
        int planned = 0;
        for (unsigned i = 0; i < n_gpu; ++i) planned += A.spmm_plan_gpu && A.spmm_plan_gpu[i] != NULL;
        printf("method %d call %d: %s, %d of %u GPUs planned, max rel err %.3g\n", method, c, o.correct ? "ok" : "MISMATCH", planned,
               n_gpu, o.max_rel);
        ok = ok && o.correct;
        // no plan on the first call (a one-shot caller pays nothing), one per GPU from the second on (unless switched off);
        // method 2 beyond 128 columns calls the per-GPU product once per 128-column tile, so its second TILE is planned
        const char *e = getenv("SBLAS_PLAN");
        const bool plans_on = !(e && e[0] == '0');
        const bool tiled = method == 2 && b_width >= 256;
        ok = ok && planned == (((c == 0 && !tiled) || !plans_on) ? 0 : (int)n_gpu);
    }
    A.sync2gpu(n_gpu, method == 1 ? replicate : segment); // a new placement drops the plans
    for (unsigned i = 0; i < n_gpu; ++i) ok = ok && A.spmm_plan_gpu[i] == NULL;
    return ok;
}

int main(int argc, char *argv[])
{
    if (argc < 4) {
        cerr << "usage: plan_test <matrix.mtx> <B_width> <gpus> [calls]" << endl;
        return 1;
    }
    const char *path = argv[1];
    const int width = atoi(argv[2]);
    const unsigned gpus = (unsigned)atoi(argv[3]);
    const int calls = argc > 4 ? atoi(argv[4]) : 3;
    bool ok = repeated(1, path, width, gpus, calls);
    ok = repeated(2, path, width, gpus, calls) && ok;
    cout << "plan_test: " << (ok ? "PASS" : "FAIL") << endl;
    return ok ? 0 : 2;
}
