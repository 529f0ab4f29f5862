/*
 * sblas_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
 *
 * A plain-C, single-threaded CPU restatement of the reference's algorithm for the
 * CSR SpMV / SpMM hot path (tartarughina/S-BLAS @ 2024-12-20).  It exists so that
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg can check (or
 * time) something that follows the reference line by line.  Nothing under
 * s-blas_amd/ may include, link or call this file.
 *
 * Parity pinning: the loader half is checked bit-for-bit against the reference's own
 * mmio.h / mmio_highlevel.h compiled from /root/reference (oracle/_ref, see
 * oracle/Makefile); the arithmetic half is pinned by the golden values the survey
 * captured from the reference's own sblas_spmm_csr_cpu / sblas_spmv_csr_cpu
 * (SURVEY.md section 8c -> tests/golden/ash85_golden.json).  spmm.h / spmv.h /
 * matrix.h themselves need cuda_runtime.h, cusparse.h and nccl.h, which this image
 * lacks, so they are "unbuildable here" and are restated, not compiled.
 *
 * Each function cites the reference file:line it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

/* ------------------------------------------------------------------------- */
/* MatrixMarket banner + size line (reference: mmio.h:254-337, :339-367)      */
/* ------------------------------------------------------------------------- */

enum { ORC_REAL = 1, ORC_COMPLEX = 2, ORC_INTEGER = 3, ORC_PATTERN = 4 };
enum { ORC_GENERAL = 1, ORC_SYMM = 2, ORC_HERM = 3, ORC_SKEW = 4 };

typedef struct {
    int field;    /* ORC_REAL ...    */
    int symmetry; /* ORC_GENERAL ... */
    int sparse;   /* coordinate = 1, array = 0 */
} orc_banner;

static void lower_inplace(char *s)
{
    for (; *s; ++s) *s = (char)tolower((unsigned char)*s);
}

/* mmio.h:254-337: first line must hold five tokens; the object, format, field and
 * symmetry tokens are lower-cased before comparison; the banner token is compared
 * with strncmp against "%%MatrixMarket".  Error codes: premature EOF = 12,
 * no header = 14, unsupported = 15 (mmio.h:122-128 numbering). */
static int read_banner(FILE *f, orc_banner *b)
{
    char line[1025];
    char tok[5][64];
    memset(b, 0, sizeof *b);
    if (!fgets(line, sizeof line, f)) return 12;
    if (sscanf(line, "%63s %63s %63s %63s %63s", tok[0], tok[1], tok[2], tok[3], tok[4]) != 5)
        return 12;
    for (int i = 1; i < 5; ++i) lower_inplace(tok[i]);
    if (strncmp(tok[0], "%%MatrixMarket", strlen("%%MatrixMarket")) != 0) return 14;
    if (strcmp(tok[1], "matrix") != 0) return 15;
    if (strcmp(tok[2], "coordinate") == 0) b->sparse = 1;
    else if (strcmp(tok[2], "array") == 0) b->sparse = 0;
    else return 15;
    if (strcmp(tok[3], "real") == 0) b->field = ORC_REAL;
    else if (strcmp(tok[3], "complex") == 0) b->field = ORC_COMPLEX;
    else if (strcmp(tok[3], "pattern") == 0) b->field = ORC_PATTERN;
    else if (strcmp(tok[3], "integer") == 0) b->field = ORC_INTEGER;
    else return 15;
    if (strcmp(tok[4], "general") == 0) b->symmetry = ORC_GENERAL;
    else if (strcmp(tok[4], "symmetric") == 0) b->symmetry = ORC_SYMM;
    else if (strcmp(tok[4], "hermitian") == 0) b->symmetry = ORC_HERM;
    else if (strcmp(tok[4], "skew-symmetric") == 0) b->symmetry = ORC_SKEW;
    else return 15;
    return 0;
}

/* mmio.h:339-367: skip every line whose first byte is '%', then the first line that
 * parses as three ints is the size line; otherwise keep fscanf-ing. */
static int read_crd_size(FILE *f, int *M, int *N, int *nz)
{
    char line[1025];
    *M = *N = *nz = 0;
    do {
        if (!fgets(line, sizeof line, f)) return 12;
    } while (line[0] == '%');
    if (sscanf(line, "%d %d %d", M, N, nz) == 3) return 0;
    for (;;) {
        int got = fscanf(f, "%d %d %d", M, N, nz);
        if (got == EOF) return 12;
        if (got == 3) return 0;
    }
}

/* The triplet pass shared by mmio_info (mmio_highlevel.h:64-90) and mmio_data
 * (:189-215).  Field precedence real > complex > integer > pattern; complex keeps the
 * real part only; pattern entries become 1.0; indices go 1-based -> 0-based. */
typedef struct {
    int m, n, nz_file, mirrored;
    int *ri, *ci;
    double *v;
} orc_triplets;

static void free_triplets(orc_triplets *t)
{
    free(t->ri); free(t->ci); free(t->v);
    memset(t, 0, sizeof *t);
}

static int read_triplets(const char *path, orc_triplets *t)
{
    memset(t, 0, sizeof *t);
    FILE *f = fopen(path, "r");
    if (!f) return -1;                              /* mmio_highlevel.h:20-21 */
    orc_banner b;
    if (read_banner(f, &b) != 0) { fclose(f); return -2; }   /* :23-26 */
    if (read_crd_size(f, &t->m, &t->n, &t->nz_file) != 0) { fclose(f); return -4; } /* :42-44 */
    /* symmetric OR hermitian are mirrored; skew-symmetric is NOT (:46-51) */
    t->mirrored = (b.symmetry == ORC_SYMM || b.symmetry == ORC_HERM);
    size_t cap = t->nz_file > 0 ? (size_t)t->nz_file : 1;
    t->ri = (int *)malloc(cap * sizeof(int));
    t->ci = (int *)malloc(cap * sizeof(int));
    t->v = (double *)malloc(cap * sizeof(double));
    for (int e = 0; e < t->nz_file; ++e) {
        int i = 0, j = 0, iv = 0;
        double re = 0.0, im = 0.0;
        if (b.field == ORC_REAL) {
            if (fscanf(f, "%d %d %lg\n", &i, &j, &re) != 3) { fclose(f); free_triplets(t); return -5; }
        } else if (b.field == ORC_COMPLEX) {
            if (fscanf(f, "%d %d %lg %lg\n", &i, &j, &re, &im) != 4) { fclose(f); free_triplets(t); return -5; }
        } else if (b.field == ORC_INTEGER) {
            if (fscanf(f, "%d %d %d\n", &i, &j, &iv) != 3) { fclose(f); free_triplets(t); return -5; }
            re = iv;
        } else {
            if (fscanf(f, "%d %d\n", &i, &j) != 2) { fclose(f); free_triplets(t); return -5; }
            re = 1.0;
        }
        t->ri[e] = i - 1;
        t->ci[e] = j - 1;
        t->v[e] = re;
    }
    fclose(f);
    return 0;
}

/* Row counts + exclusive scan (mmio_highlevel.h:82,95-112 / :207,220-236). */
static void count_and_scan(const orc_triplets *t, int *rowptr)
{
    memset(rowptr, 0, (size_t)(t->m + 1) * sizeof(int));
    for (int e = 0; e < t->nz_file; ++e) rowptr[t->ri[e]]++;
    if (t->mirrored)
        for (int e = 0; e < t->nz_file; ++e)
            if (t->ri[e] != t->ci[e]) rowptr[t->ci[e]]++;
    int carry = rowptr[0];
    rowptr[0] = 0;
    for (int r = 1; r <= t->m; ++r) {
        int here = rowptr[r];
        rowptr[r] = carry + rowptr[r - 1];
        carry = here;
    }
}

/* mmio_info, mmio_highlevel.h:7-127. */
int orc_mm_info(const char *path, int *m, int *n, int *nnz, int *is_symmetric)
{
    orc_triplets t;
    int rc = read_triplets(path, &t);
    if (rc) return rc;
    int *rowptr = (int *)malloc((size_t)(t.m + 1) * sizeof(int));
    count_and_scan(&t, rowptr);
    *m = t.m; *n = t.n; *nnz = rowptr[t.m]; *is_symmetric = t.mirrored;
    free(rowptr);
    free_triplets(&t);
    return 0;
}

/* mmio_data, mmio_highlevel.h:130-281: scatter in FILE ORDER with one cursor per row;
 * for a mirrored file the (i,j) copy is placed first, then the (j,i) copy (:242-262).
 * Rows are NOT sorted afterwards. */
int orc_mm_data(const char *path, int *rowptr, int *colidx, double *val)
{
    orc_triplets t;
    int rc = read_triplets(path, &t);
    if (rc) return rc;
    count_and_scan(&t, rowptr);
    int *fill = (int *)calloc((size_t)(t.m + 1), sizeof(int));
    for (int e = 0; e < t.nz_file; ++e) {
        int i = t.ri[e], j = t.ci[e];
        int at = rowptr[i] + fill[i]++;
        colidx[at] = j;
        val[at] = t.v[e];
        if (t.mirrored && i != j) {
            at = rowptr[j] + fill[j]++;
            colidx[at] = i;
            val[at] = t.v[e];
        }
    }
    free(fill);
    free_triplets(&t);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Arithmetic: the reference's CPU verifier                                   */
/* ------------------------------------------------------------------------- */

/* sblas_spmm_csr_cpu, column-major C branch, spmm.h:56-68.
 * Loop order i -> n -> j; sum starts at 0; C = beta*C + alpha*sum.
 * B is K x N column-major (ld = K, spmm.h:62), C is M x N column-major (ld = M, :65-66). */
void orc_spmm_csr(int M, int K, int N, const int *rowptr, const int *colidx,
                  const double *val, const double *B, double *C, double alpha, double beta)
{
    for (int i = 0; i < M; ++i) {
        for (int n = 0; n < N; ++n) {
            double sum = 0;
            for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) {
                int col = colidx[j];
                double a = val[j];
                double b = B[(size_t)n * (size_t)K + (size_t)col];
                sum += a * b;
            }
            size_t at = (size_t)n * (size_t)M + (size_t)i;
            C[at] = beta * C[at] + alpha * sum;
        }
    }
}

/* Same arithmetic on a row range [r0, r1) -- used by bench.py's bounded cpu_baseline
 * sample and by the tests' emulation of the g-way row partition. */
void orc_spmm_csr_rows(int r0, int r1, int M, int K, int N, const int *rowptr,
                       const int *colidx, const double *val, const double *B, double *C,
                       double alpha, double beta)
{
    for (int i = r0; i < r1; ++i) {
        for (int n = 0; n < N; ++n) {
            double sum = 0;
            for (int j = rowptr[i]; j < rowptr[i + 1]; ++j)
                sum += val[j] * B[(size_t)n * (size_t)K + (size_t)colidx[j]];
            size_t at = (size_t)n * (size_t)M + (size_t)i;
            C[at] = beta * C[at] + alpha * sum;
        }
    }
}

/* The same loop with the rows spread over OpenMP threads (rows are independent): bench.py's "all host cores"
 * figure.  The reference itself is single-threaded; arithmetic per element is unchanged. */
void orc_spmm_csr_omp(int M, int K, int N, const int *rowptr, const int *colidx, const double *val,
                      const double *B, double *C, double alpha, double beta)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < M; ++i) {
        for (int n = 0; n < N; ++n) {
            double sum = 0;
            for (int j = rowptr[i]; j < rowptr[i + 1]; ++j)
                sum += val[j] * B[(size_t)n * (size_t)K + (size_t)colidx[j]];
            size_t at = (size_t)n * (size_t)M + (size_t)i;
            C[at] = beta * C[at] + alpha * sum;
        }
    }
}

/* sblas_spmv_csr_cpu, spmv.h:22-31. */
void orc_spmv_csr(int M, const int *rowptr, const int *colidx, const double *val,
                  const double *x, double *y, double alpha, double beta)
{
    for (int i = 0; i < M; ++i) {
        double sum = 0;
        for (int j = rowptr[i]; j < rowptr[i + 1]; ++j)
            sum += val[j] * x[colidx[j]];
        y[i] = beta * y[i] + alpha * sum;
    }
}

/* The same two loops in the other instantiations of the reference's templates -- sblas_spmm_csr_cpu<IdxType, DataType>
 * (spmm.h:29-68) and sblas_spmv_csr_cpu<IdxType, DataType> (spmv.h:15-31) with DataType float and / or IdxType
 * int64_t (utility.h:302-316 lists the types the GPU paths map).  `DataType sum = 0` accumulates in the value type. */
#define ORC_TYPED(NAME, I, T)                                                                                   \
    void orc_spmm_csr_##NAME(int64_t M, int64_t K, int64_t N, const I *rowptr, const I *colidx, const T *val,   \
                             const T *B, T *C, T alpha, T beta)                                                \
    {                                                                                                           \
        for (int64_t i = 0; i < M; ++i)                                                                         \
            for (int64_t n = 0; n < N; ++n) {                                                                   \
                T sum = 0;                                                                                      \
                for (I j = rowptr[i]; j < rowptr[i + 1]; ++j) sum += val[j] * B[(size_t)n * (size_t)K + (size_t)colidx[j]]; \
                size_t at = (size_t)n * (size_t)M + (size_t)i;                                                  \
                C[at] = beta * C[at] + alpha * sum;                                                             \
            }                                                                                                   \
    }                                                                                                           \
    void orc_spmv_csr_##NAME(int64_t M, const I *rowptr, const I *colidx, const T *val, const T *x, T *y,       \
                             T alpha, T beta)                                                                   \
    {                                                                                                           \
        for (int64_t i = 0; i < M; ++i) {                                                                       \
            T sum = 0;                                                                                          \
            for (I j = rowptr[i]; j < rowptr[i + 1]; ++j) sum += val[j] * x[colidx[j]];                         \
            y[i] = beta * y[i] + alpha * sum;                                                                   \
        }                                                                                                       \
    }
ORC_TYPED(f32_i32, int32_t, float)
ORC_TYPED(f64_i64, int64_t, double)
ORC_TYPED(f32_i64, int64_t, float)

/* denseVector_plusEqual_denseVector, kernel.h:27-38: y = y*beta + x*alpha. */
void orc_axpby(size_t n, double alpha, const double *x, double beta, double *y)
{
    for (size_t i = 0; i < n; ++i) y[i] = y[i] * beta + x[i] * alpha;
}

/* DenseMatrix(h, w, order) random fill: matrix.h:519-528 + utility.h:197 + config.h:23.
 * srand(211) then rand()/RAND_MAX in linear storage order (glibc rand()). */
void orc_fill_rand0to1(double *v, size_t n)
{
    srand(211);
    for (size_t i = 0; i < n; ++i) v[i] = (double)rand() / (double)RAND_MAX;
}

/* check_equal, utility.h:182-193: absolute tolerance ERROR_BAR = 1e-3 (config.h:21).
 * Returns 1 when every |x-y| <= 1e-3. */
int orc_check_equal(const double *x, const double *y, size_t m)
{
    int ok = 1;
    for (size_t i = 0; i < m; ++i)
        if (fabs(x[i] - y[i]) > 1e-3) ok = 0;
    return ok;
}

/* ------------------------------------------------------------------------- */
/* Multi-GPU placement arithmetic                                             */
/* ------------------------------------------------------------------------- */

/* csr_findRowIdxUsingNnzIdx, utility.h:292-300: first row r with
 * rowptr[r] <= idx < rowptr[r+1] (empty rows are skipped); -1 when none. */
int orc_find_row(const int *rowptr, int height, int nnz_idx)
{
    for (int r = 0; r < height; ++r)
        if (rowptr[r] <= nnz_idx && nnz_idx < rowptr[r + 1]) return r;
    return -1;
}

/* CsrSparseMatrix::sync2gpu(segment), matrix.h:356-375.
 * avg = ceil((float)nnz / g)  -- the reference's single-precision ceil, kept here on
 * purpose (matrix.h:360); see orc_avg_nnz_exact for the integer form the product uses.
 * rebased (length stop-start+2): [0, rowptr[start+k]-i*avg ..., nnz_i] (:370-375). */
int orc_avg_nnz_float(int nnz, int g) { return (int)ceilf((float)nnz / (float)g); }
int orc_avg_nnz_exact(int nnz, int g) { return (int)(((long long)nnz + g - 1) / g); }

int orc_partition_nnz(const int *rowptr, int M, int nnz, int g, int i, int avg,
                      int *start_row, int *stop_row, int *nnz_i, int *rebased)
{
    (void)g;
    long long first = (long long)i * avg;
    long long lastp1 = (long long)(i + 1) * avg;
    if (lastp1 > nnz) lastp1 = nnz;
    int last = (int)lastp1 - 1;
    *nnz_i = last - (int)first + 1;
    *start_row = orc_find_row(rowptr, M, (int)first);
    *stop_row = orc_find_row(rowptr, M, last);
    if (*start_row < 0 || *stop_row < 0) return -1;
    int num = *stop_row - *start_row + 2;            /* matrix.h:398-404 */
    if (rebased) {
        rebased[0] = 0;
        for (int k = 1; k < num - 1; ++k) rebased[k] = rowptr[*start_row + k] - (int)first;
        rebased[num - 1] = *nnz_i;
    }
    return num;
}

/* DenseMatrix::sync2gpu(segment), matrix.h:554-568: avg = ceil((double)first/g),
 * dim_i = min((i+1)*avg, first) - i*avg, block offset = i*avg (in leading-dim units). */
void orc_partition_dense(int first_order, int g, int i, int *offset, int *dim)
{
    int avg = (int)ceil((double)first_order / (double)g);
    int hi = (i + 1) * avg;
    if (hi > first_order) hi = first_order;
    *offset = i * avg;
    *dim = hi - i * avg;
}

/* ------------------------------------------------------------------------- */
/* Fingerprints used by the golden fixtures                    */
/* ------------------------------------------------------------------------- */
uint64_t orc_fnv1a64(const void *data, size_t nbytes)
{
    const unsigned char *p = (const unsigned char *)data;
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < nbytes; ++i) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}
