// pipeline_test -- method 2 with and without the column-tile pipeline (spmm.h: SpMM of tile c + 1 beside the exchange
// and scatter of tile c): the two results must be identical bit for bit (same kernels, same order of the terms), and
// both must match the host verifier.   pipeline_test <matrix> <B_width> <gpus>
#include <cstring>

#include "harness.h"

static bool run(const char *path, int b_width, unsigned n_gpu, const char *pipeline, std::vector<double> &out)
{
    setenv("SBLAS_M2_PIPELINE", pipeline, 1);
    CsrSparseMatrix<int, double> A(path);
    if (A.height == 0 || A.nnz == 0) return false;
    DenseMatrix<int, double> B(A.width, b_width, col_major);
    DenseMatrix<int, double> C(A.height, b_width, 1.0, col_major), C_cpu(A.height, b_width, 1.0, col_major);
    A.sync2gpu(n_gpu, segment);
    B.sync2gpu(n_gpu, replicate);
    C.sync2gpu(n_gpu, replicate);
    bool ok = true;
    for (int call = 0; call < 2; ++call) { // the second call runs on per-GPU plans
        sblas_spmm_csr_v2<int, double>(&A, &B, &C, 3.0, 0.5, n_gpu);
        CUDA_CHECK_ERROR();
        sblas_spmm_csr_cpu<int, double>(&A, &B, &C_cpu, 3.0, 0.5);
    }
    for (unsigned i = 0; i < n_gpu; ++i) { // every GPU holds the full result
        C.sync2cpu(i);
        const harness::Outcome o = harness::compare(C_cpu.val, C.val, C.get_mtx_num());
        printf("pipeline %s, GPU %u: %s (max rel err %.3g)\n", pipeline, i, o.correct ? "ok" : "MISMATCH", o.max_rel);
        ok = ok && o.correct;
    }
    out.assign(C.val, C.val + C.get_mtx_num());
    return ok;
}

int main(int argc, char *argv[])
{
    if (argc < 4) {
        cerr << "usage: pipeline_test <matrix.mtx> <B_width> <gpus>" << endl;
        return 1;
    }
    std::vector<double> serial, piped;
    bool ok = run(argv[1], atoi(argv[2]), (unsigned)atoi(argv[3]), "0", serial);
    ok = run(argv[1], atoi(argv[2]), (unsigned)atoi(argv[3]), "1", piped) && ok;
    const bool same = serial.size() == piped.size() && memcmp(serial.data(), piped.data(), serial.size() * sizeof(double)) == 0;
    cout << "bit-identical: " << (same ? "yes" : "NO") << endl;
    cout << "pipeline_test: " << (ok && same ? "PASS" : "FAIL") << endl;
    return ok && same ? 0 : 2;
}
