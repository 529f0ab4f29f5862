// unit_test -- container smoke tests plus the three ops on ./ash85.mtx with alpha = 3, beta = 4, N = 256 and
// 4 (logical) GPUs, as the reference's unit_test.cu:177-187.  Optional arguments: [matrix_path [gpus]].
// Exit status is non-zero when any validation fails.
#include "harness.h"

static bool containers(const char *path)
{
    bool ok = true;
    CooSparseMatrix<int, double> coo_empty;
    CooSparseMatrix<int, double> coo(path);
    for (int k = 1; k < coo.nnz; ++k) // sorted by (row, col)
        ok &= (coo.cooRowIdx[k - 1] < coo.cooRowIdx[k]) ||
              (coo.cooRowIdx[k - 1] == coo.cooRowIdx[k] && coo.cooColIdx[k - 1] <= coo.cooColIdx[k]);
    CsrSparseMatrix<unsigned, double> csr_empty;
    CsrSparseMatrix<unsigned, double> csr(path);
    ok &= csr.csrRowPtr[csr.height] == csr.nnz;
    CscSparseMatrix<unsigned, double> csc_empty;
    CscSparseMatrix<unsigned, double> csc(&csr);
    ok &= csc.cscColPtr[csc.width] == csc.nnz;
    // CSR -> CSC -> CSR is the identity when rows are ascending (ash85's are)
    std::vector<unsigned> rp(csr.height + 1), ci(csr.nnz);
    std::vector<double> v(csr.nnz);
    CscToCsr<unsigned, double>(csc.width, csc.height, csc.nnz, csc.cscColPtr, csc.cscRowIdx, csc.cscVal, ci.data(),
                               rp.data(), v.data());
    for (unsigned i = 0; i <= csr.height; ++i) ok &= rp[i] == csr.csrRowPtr[i];
    for (unsigned k = 0; k < csr.nnz; ++k) ok &= ci[k] == csr.csrColIdx[k] && v[k] == csr.csrVal[k];
    DenseMatrix<unsigned, double> dm_empty;
    DenseMatrix<unsigned, double> dm(257, 129, row_major);
    DenseMatrix<unsigned, double> *t = dm.transpose();
    ok &= t->order == col_major && t->val[5 * 257 + 7] == dm.val[7 * 129 + 5];
    delete t;
    cout << "Containers = " << (ok ? "True" : "False") << endl;
    return ok;
}

int main(int argc, char *argv[])
{
    const char *path = argc > 1 ? argv[1] : "./ash85.mtx";
    const unsigned gpus = argc > 2 ? (unsigned)atoi(argv[2]) : 4;
    bool ok = containers(path);
    ok &= harness::spmm(1, path, 256, 3.0, 4.0, gpus);
    ok &= harness::spmm(2, path, 256, 3.0, 4.0, gpus);
    ok &= harness::spmv(path, 3.0, 4.0, gpus);
    cout << "unit_test: " << (ok ? "PASS" : "FAIL") << endl;
    return ok ? 0 : 2;
}
