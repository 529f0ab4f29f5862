// typed_test -- the three ops in the value / index types besides <int, double>: the reference's templates accept
// float or double values and 32- or 64-bit indices (utility.h:302-316) although its own drivers instantiate
// <int, double> only.  For each of <int, float>, <int64_t, double>, <int64_t, float>: SpMM method 1, SpMM method 2
// (both merges), SpMV, each against the host loop of the same types.
//   typed_test [matrix_path [gpus [b_width]]]        default ./ash85.mtx 4 100
// Bars: fp64 1e-10 relative (as the other drivers), fp32 1e-4 relative to max(1, |ref|) (sums of up to a few hundred
// fp32 terms, fused multiply-add on the GPU against multiply + add on the host).  Exit status non-zero on a failure.
#include <cmath>
#include <string>

#include "matrix.h"
#include "sblas.h"

template <typename T> static double bar() { return sizeof(T) == 4 ? 1e-4 : 1e-10; }

template <typename T> static bool close_enough(const char *what, const T *ref, const T *got, size_t n)
{
    double worst = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const double d = std::fabs((double)ref[i] - (double)got[i]) / std::fmax(1.0, std::fabs((double)ref[i]));
        if (!(d <= worst)) worst = d; // NaN sticks
    }
    const bool ok = worst <= bar<T>();
    cout << what << ": max relative error " << worst << (ok ? "  ok" : "  FAIL") << endl;
    return ok;
}

template <typename I, typename T> static bool run_types(const char *name, const char *path, unsigned gpus, int b_width)
{
    cout << "---- " << name << " ----" << endl;
    const T alpha = (T)3, beta = (T)4;
    bool ok = true;
    for (int method = 1; method <= 2; ++method) {
        for (int merge = 0; merge < (method == 2 ? 2 : 1); ++merge) {
            if (method == 2) setenv("SBLAS_MERGE", merge ? "allreduce" : "rowblocks", 1);
            CsrSparseMatrix<I, T> A(path);
            if (A.height == 0 || A.nnz == 0) {
                cerr << "empty or unreadable matrix: " << path << endl;
                return false;
            }
            DenseMatrix<I, T> B(A.width, (I)b_width, col_major);
            DenseMatrix<I, T> C(A.height, (I)b_width, (T)1, col_major);
            DenseMatrix<I, T> C_cpu(A.height, (I)b_width, (T)1, col_major);
            A.sync2gpu(gpus, method == 1 ? replicate : segment);
            B.sync2gpu(gpus, method == 1 ? segment : replicate);
            C.sync2gpu(gpus, method == 1 ? segment : replicate);
            if (method == 1) sblas_spmm_csr_v1<I, T>(&A, &B, &C, alpha, beta, gpus);
            else sblas_spmm_csr_v2<I, T>(&A, &B, &C, alpha, beta, gpus);
            CUDA_CHECK_ERROR();
            if (method == 2) C.sync2cpu(gpus - 1);
            sblas_spmm_csr_cpu<I, T>(&A, &B, &C_cpu, alpha, beta);
            const std::string what = std::string("spmm method ") + (method == 1 ? "1" : merge ? "2 (all-reduce)" : "2 (row blocks)");
            ok &= close_enough<T>(what.c_str(), C_cpu.val, C.val, C.get_mtx_num());
        }
    }
    for (int merge = 0; merge < 2; ++merge) {
        setenv("SBLAS_MERGE", merge ? "allreduce" : "rowblocks", 1);
        CsrSparseMatrix<I, T> A(path);
        DenseVector<I, T> x(A.width, (T)1);
        DenseVector<I, T> y(A.height, (T)1);
        DenseVector<I, T> y_cpu(A.height, (T)1);
        for (size_t i = 0; i < (size_t)A.width; ++i) x.val[i] = (T)(0.25 + (double)(i % 7) / 8.0);
        A.sync2gpu(gpus, segment);
        x.sync2gpu(gpus, replicate);
        y.sync2gpu(gpus, replicate);
        sblas_spmv_csr_v1<I, T>(&A, &x, &y, alpha, beta, gpus);
        CUDA_CHECK_ERROR();
        y.sync2cpu(0);
        sblas_spmv_csr_cpu<I, T>(&A, &x, &y_cpu, alpha, beta);
        ok &= close_enough<T>(merge ? "spmv (all-reduce)" : "spmv (row blocks)", y_cpu.val, y.val, (size_t)A.height);
    }
    unsetenv("SBLAS_MERGE");
    return ok;
}

int main(int argc, char *argv[])
{
    const char *path = argc > 1 ? argv[1] : "./ash85.mtx";
    const unsigned gpus = argc > 2 ? (unsigned)atoi(argv[2]) : 4;
    const int b_width = argc > 3 ? atoi(argv[3]) : 100;
    if (gpus == 0) {
        cerr << "typed_test needs at least one GPU" << endl;
        return 2;
    }
    bool ok = true;
    ok &= run_types<int, float>("<int, float>", path, gpus, b_width);
    ok &= run_types<int64_t, double>("<int64_t, double>", path, gpus, b_width);
    ok &= run_types<int64_t, float>("<int64_t, float>", path, gpus, b_width);
    ok &= run_types<int, double>("<int, double> (the tuned path, through the typed entry points)", path, gpus, b_width);
    cout << "typed_test: " << (ok ? "PASS" : "FAIL") << endl;
    return ok ? 0 : 2;
}
