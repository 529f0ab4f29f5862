// harness.h -- shared test harness of the three command-line drivers.
// Same flow and the same four report lines as the reference's drivers (spmm_test.cu:19-93, spmv_test.cu:6-42):
// build A / B / C, place them on the GPUs, run the GPU op, run the host verifier, compare with check_equal.
// Added: gpus = 0 runs the host path only (BASELINE config 1), the comparison also reports the largest
// relative error, and a mismatch makes the process exit non-zero (the reference always exits 0).
#ifndef SBLAS_AMD_DRIVER_HARNESS_H
#define SBLAS_AMD_DRIVER_HARNESS_H

#include <cmath>
#include <string>

#include "matrix.h"
#include "sblas.h"

namespace harness {

struct Outcome {
    bool correct = true;
    double max_rel = 0.0;
};

inline Outcome compare(const double *ref, const double *got, size_t n)
{
    Outcome o;
    o.correct = check_equal(ref, got, n);
    for (size_t i = 0; i < n; ++i) {
        const double d = std::fabs(ref[i] - got[i]) / std::fmax(1.0, std::fabs(ref[i]));
        if (!(d <= o.max_rel)) o.max_rel = d; // NaN sticks
    }
    if (!(o.max_rel <= 1e-10)) o.correct = false; // this build's bar: fp64 within 1e-10 relative
    return o;
}

inline void report(const Outcome &o, double load_ms, double gpu_ms, double cpu_ms, unsigned n_gpu, bool cpu_first)
{
    cout << "Validation = " << (o.correct ? "True" : "False") << endl;
    cout << "Load Time: " << load_ms << "ms." << endl;
    if (cpu_first) cout << "CPU Run Time: " << cpu_ms << " ms." << endl;
    cout << n_gpu << "-GPUs Run Time: " << gpu_ms << " ms." << endl;
    if (!cpu_first) cout << "CPU Run Time: " << cpu_ms << " ms." << endl;
    cout << "Max relative error vs CPU verifier: " << o.max_rel << endl;
}

// method 1 = partition B/C by columns, method 2 = partition A by nonzeros
inline bool spmm(int method, const char *path, int b_width, double alpha, double beta, unsigned n_gpu)
{
    cpu_timer t_load, t_gpu, t_cpu;
    t_load.start_timer();
    CsrSparseMatrix<int, double> A(path);
    if (A.height == 0 || A.nnz == 0) {
        cerr << "empty or unreadable matrix: " << path << endl;
        return false;
    }
    DenseMatrix<int, double> B(A.width, b_width, col_major);
    DenseMatrix<int, double> C(A.height, b_width, 1.0, col_major);
    DenseMatrix<int, double> C_cpu(A.height, b_width, 1.0, col_major);
    if (n_gpu > 0) {
        A.sync2gpu(n_gpu, method == 1 ? replicate : segment);
        B.sync2gpu(n_gpu, method == 1 ? segment : replicate);
        C.sync2gpu(n_gpu, method == 1 ? segment : replicate);
        CUDA_SAFE_CALL(cudaDeviceSynchronize());
    }
    t_load.stop_timer();

    if (n_gpu > 0) {
        t_gpu.start_timer();
        if (method == 1) sblas_spmm_csr_v1<int, double>(&A, &B, &C, alpha, beta, n_gpu);
        else sblas_spmm_csr_v2<int, double>(&A, &B, &C, alpha, beta, n_gpu);
        CUDA_CHECK_ERROR();
        t_gpu.stop_timer();
    }
    t_cpu.start_timer();
    sblas_spmm_csr_cpu<int, double>(&A, &B, &C_cpu, alpha, beta);
    t_cpu.stop_timer();

    Outcome o;
    if (n_gpu > 0) {
        if (method == 2) C.sync2cpu(0); // every GPU holds the full result; method 1 already gathered it
        o = compare(C_cpu.val, C.val, C.get_mtx_num());
    } else {
        cout << "gpus = 0: host path only" << endl;
        double sum = 0.0;
        for (size_t i = 0; i < C_cpu.get_mtx_num(); ++i) sum += C_cpu.val[i];
        printf("C[0] = %.17g  C[last] = %.17g  sum(C) = %.17g\n", C_cpu.val[0], C_cpu.val[C_cpu.get_mtx_num() - 1], sum);
    }
    report(o, t_load.measure(), t_gpu.measure(), t_cpu.measure(), n_gpu, false);
    const double gflop = 2.0 * (double)A.nnz * b_width * 1e-9;
    if (n_gpu > 0) printf("GPU op: %.3f GFLOP/s (whole call, incl. per-call setup and copies)\n", gflop / (t_gpu.measure() * 1e-3));
    printf("CPU verifier: %.3f GFLOP/s (1 thread)\n", gflop / (t_cpu.measure() * 1e-3));
    return o.correct;
}

inline bool spmv(const char *path, double alpha, double beta, unsigned n_gpu)
{
    cpu_timer t_load, t_gpu, t_cpu;
    t_load.start_timer();
    CsrSparseMatrix<int, double> A(path);
    if (A.height == 0 || A.nnz == 0) {
        cerr << "empty or unreadable matrix: " << path << endl;
        return false;
    }
    DenseVector<int, double> x(A.width, 1.);
    DenseVector<int, double> y(A.height, 1.);
    DenseVector<int, double> y_cpu(A.height, 1.);
    if (n_gpu > 0) {
        A.sync2gpu(n_gpu, segment);
        x.sync2gpu(n_gpu, replicate);
        y.sync2gpu(n_gpu, replicate);
        CUDA_SAFE_CALL(cudaDeviceSynchronize());
    }
    t_load.stop_timer();
    t_cpu.start_timer();
    sblas_spmv_csr_cpu<int, double>(&A, &x, &y_cpu, alpha, beta);
    t_cpu.stop_timer();
    Outcome o;
    if (n_gpu > 0) {
        t_gpu.start_timer();
        sblas_spmv_csr_v1<int, double>(&A, &x, &y, alpha, beta, n_gpu);
        CUDA_CHECK_ERROR();
        t_gpu.stop_timer();
        y.sync2cpu(0);
        o = compare(y_cpu.val, y.val, y.get_vec_length());
    } else {
        cout << "gpus = 0: host path only" << endl;
    }
    report(o, t_load.measure(), t_gpu.measure(), t_cpu.measure(), n_gpu, true);
    return o.correct;
}

} // namespace harness
#endif
