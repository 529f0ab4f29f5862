// host_util.cpp -- host-side pure functions of the C ABI: placement arithmetic and the
// MatrixMarket -> CSR loader.  No GPU call in this file; everything is testable on a CPU box.
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <sys/stat.h>
#include <atomic>
#include <thread>
#include <chrono>
#include <vector>
#include "../../include/sblas_hip.h"

// ------------------------------------------------------------------------------------------
// placement
// ------------------------------------------------------------------------------------------
extern "C" int32_t sblas_find_row_of_nnz(const int32_t *rowptr, int32_t rows, int32_t nnz_idx)
{
    // The reference scans linearly for the first r with rowptr[r] <= idx < rowptr[r+1]
    // (utility.h:292-300).  rowptr is non-decreasing, so that r is the predecessor of the first
    // entry strictly greater than idx.
    if (!rowptr || rows <= 0 || nnz_idx < rowptr[0] || nnz_idx >= rowptr[rows]) return -1;
    const int32_t *hi = std::upper_bound(rowptr, rowptr + rows + 1, nnz_idx);
    return (int32_t)(hi - rowptr) - 1;
}

extern "C" int64_t sblas_partition_nnz(const int32_t *rowptr, int32_t rows, int32_t nnz, int n_gpu, int i_gpu,
                                       int32_t *start_row, int32_t *stop_row, int32_t *nnz_i,
                                       int64_t *first_nnz, int32_t *rebased_rowptr)
{
    if (!rowptr || rows <= 0 || nnz <= 0 || n_gpu <= 0 || i_gpu < 0 || i_gpu >= n_gpu) return -1;
    // matrix.h:360 computes ceil((float)nnz / n_gpu); float loses integers above 2^24, so the
    // product uses the exact quotient (identical below 2^24, see DESIGN.md "hazards").
    const int64_t avg = ((int64_t)nnz + n_gpu - 1) / n_gpu;
    const int64_t lo = (int64_t)i_gpu * avg;
    const int64_t hi = std::min<int64_t>((int64_t)(i_gpu + 1) * avg, nnz); // one past the last
    if (lo >= hi) return -2; // this rank owns nothing (g larger than the matrix can feed)
    const int32_t s = sblas_find_row_of_nnz(rowptr, rows, (int32_t)lo);
    const int32_t e = sblas_find_row_of_nnz(rowptr, rows, (int32_t)(hi - 1));
    if (s < 0 || e < 0) return -3;
    if (start_row) *start_row = s;
    if (stop_row) *stop_row = e;
    if (nnz_i) *nnz_i = (int32_t)(hi - lo);
    if (first_nnz) *first_nnz = lo;
    const int64_t num = (int64_t)e - s + 2; // get_gpu_row_ptr_num, matrix.h:398-404
    if (rebased_rowptr) {
        // matrix.h:370-375: first pointer 0, interior pointers shifted by the block's first
        // nonzero, last pointer = the block's nonzero count (a row cut by the boundary is split).
        rebased_rowptr[0] = 0;
        for (int64_t k = 1; k < num - 1; ++k) rebased_rowptr[k] = (int32_t)(rowptr[s + k] - lo);
        rebased_rowptr[num - 1] = (int32_t)(hi - lo);
    }
    return num;
}

// The same partition for CsrSparseMatrix<int64_t, T> (the reference's sync2gpu is one template, matrix.h:356-375).
extern "C" int64_t sblas_partition_nnz_i64(const int64_t *rowptr, int64_t rows, int64_t nnz, int n_gpu, int i_gpu,
                                           int64_t *start_row, int64_t *stop_row, int64_t *nnz_i, int64_t *first_nnz,
                                           int64_t *rebased_rowptr)
{
    if (!rowptr || rows <= 0 || nnz <= 0 || n_gpu <= 0 || i_gpu < 0 || i_gpu >= n_gpu) return -1;
    const int64_t avg = (nnz + n_gpu - 1) / n_gpu;
    const int64_t lo = (int64_t)i_gpu * avg;
    const int64_t hi = std::min<int64_t>((int64_t)(i_gpu + 1) * avg, nnz);
    if (lo >= hi) return -2;
    // row of a nonzero: the last row whose first nonzero is not beyond it (utility.h:292-300)
    auto row_of = [&](int64_t k) { return (int64_t)(std::upper_bound(rowptr, rowptr + rows + 1, k) - rowptr) - 1; };
    const int64_t s = row_of(lo), e = row_of(hi - 1);
    if (s < 0 || e < 0 || s >= rows || e >= rows) return -3;
    if (start_row) *start_row = s;
    if (stop_row) *stop_row = e;
    if (nnz_i) *nnz_i = hi - lo;
    if (first_nnz) *first_nnz = lo;
    const int64_t num = e - s + 2;
    if (rebased_rowptr) {
        rebased_rowptr[0] = 0;
        for (int64_t k = 1; k < num - 1; ++k) rebased_rowptr[k] = rowptr[s + k] - lo;
        rebased_rowptr[num - 1] = hi - lo;
    }
    return num;
}

// The dense initialiser of the reference's DenseMatrix / DenseVector constructors (matrix.h:519-528, :663-672):
// srand(seed) and rand() / RAND_MAX in storage order -- the C library's generator, so callers outside C++ (bench.py)
// can build the very B the reference's drivers multiply.  Not re-entrant (global libc state), like the reference.
extern "C" int sblas_host_fill_rand0to1(double *dst, int64_t count, unsigned seed)
{
    if (count < 0 || (count > 0 && !dst)) return SBLAS_E_INVALID;
    srand(seed);
    for (int64_t i = 0; i < count; ++i) dst[i] = (double)rand() / (double)RAND_MAX;
    return SBLAS_OK;
}

extern "C" int sblas_partition_dense(int64_t first_order, int n_gpu, int i_gpu, int64_t *offset, int64_t *dim)
{
    if (first_order < 0 || n_gpu <= 0 || i_gpu < 0 || i_gpu >= n_gpu || !offset || !dim) return SBLAS_E_INVALID;
    const int64_t avg = (first_order + n_gpu - 1) / n_gpu; // == ceil((double)first/g), matrix.h:559
    const int64_t lo = (int64_t)i_gpu * avg;
    const int64_t hi = std::min<int64_t>(lo + avg, first_order);
    // the reference lets dim go negative when g*avg - avg > first (e.g. 9 columns on 8 GPUs);
    // here such a rank simply owns an empty block.
    *offset = std::min<int64_t>(lo, first_order);
    *dim = hi > lo ? hi - lo : 0;
    return SBLAS_OK;
}

// ------------------------------------------------------------------------------------------
// MatrixMarket loader: one read of the file, hand-rolled tokenizer, then the same
// count -> exclusive scan -> scatter-in-file-order construction as mmio_highlevel.h:130-281.
// ------------------------------------------------------------------------------------------
namespace {

struct Parsed {
    std::string path;
    long long mtime_ns = 0;
    long long size = -1;
    int32_t rows = 0, cols = 0, nnz = 0, mirrored = 0;
    std::vector<int32_t> rowptr, colidx;
    std::vector<double> val;
};

std::mutex g_cache_mu;
Parsed g_cache; // last file parsed: sblas_mm_read_info + sblas_mm_read_csr share one parse

enum Field { F_REAL, F_COMPLEX, F_INTEGER, F_PATTERN };

inline const char *skip_ws(const char *p, const char *end)
{
    while (p < end && isspace((unsigned char)*p)) ++p;
    return p;
}
inline const char *token_end(const char *p, const char *end)
{
    while (p < end && !isspace((unsigned char)*p)) ++p;
    return p;
}
inline std::string lowered(const char *b, const char *e)
{
    std::string s(b, e);
    for (auto &ch : s) ch = (char)tolower((unsigned char)ch);
    return s;
}
// "%d"-like: optional sign + digits; false when no digits were consumed
inline bool parse_int(const char *&p, const char *end, long &out)
{
    p = skip_ws(p, end);
    const char *q = p;
    bool neg = false;
    if (q < end && (*q == '-' || *q == '+')) neg = (*q++ == '-');
    if (q >= end || !isdigit((unsigned char)*q)) return false;
    long v = 0;
    while (q < end && isdigit((unsigned char)*q)) v = v * 10 + (*q++ - '0');
    out = neg ? -v : v;
    p = q;
    return true;
}
// "%lg": glibc's scanf hands the token to the same conversion as strtod
inline bool parse_real(const char *&p, const char *end, double &out)
{
    p = skip_ws(p, end);
    if (p >= end) return false;
    char *stop = nullptr;
    out = strtod(p, &stop); // buffer is NUL-terminated by the caller
    if (stop == p) return false;
    p = stop;
    return true;
}

// one entry = tpe whitespace-separated tokens: "i j [re [im]]".  STRICT (the threaded tokenizer): every field must end at
// whitespace or at the end of the buffer -- fscanf semantics let a conversion stop INSIDE a token ("1+2 3.0" is one
// entry for the sequential loop but two whitespace tokens for the chunk arithmetic), and such a file must go to the
// sequential loop, which mirrors the reference, instead of being realigned at a chunk boundary.
template <bool STRICT = false>
inline bool parse_entry(const char *&p, const char *end, Field field, long M, long N, bool mirrored, int32_t &ri,
                        int32_t &ci, double &v)
{
    long i = 0, j = 0;
    double re = 1.0;
    auto whole = [&]() { return !STRICT || p >= end || isspace((unsigned char)*p); };
    if (!parse_int(p, end, i) || !whole() || !parse_int(p, end, j) || !whole()) return false;
    if (field == F_REAL) {
        if (!parse_real(p, end, re) || !whole()) return false;
    } else if (field == F_COMPLEX) {
        double im;
        if (!parse_real(p, end, re) || !whole() || !parse_real(p, end, im) || !whole()) return false; // imaginary part dropped
    } else if (field == F_INTEGER) {
        long iv;
        if (!parse_int(p, end, iv) || !whole()) return false;
        re = (double)(int)iv;
    }
    if (i < 1 || i > M || j < 1 || j > N) return false; // the reference would write out of bounds
    if (mirrored && i != j && (j > M || i > N)) return false;
    ri = (int32_t)(i - 1);
    ci = (int32_t)(j - 1);
    v = re;
    return true;
}

// The entry list of a large file, tokenised by several threads (the first, uncached parse of a
// multi-GB file).  The list is a token stream (the reference reads it with fscanf, so line breaks mean nothing): the
// byte range is cut into chunks at whitespace, a first pass counts the tokens of every chunk, the prefix sum tells
// every chunk which of its tokens starts an entry, and each thread then converts its entries -- running past its
// chunk's end to finish the last one -- into their final positions: file order is preserved exactly.  Any
// irregularity (a token that does not parse, an index out of range, too few tokens) makes the caller fall back to
// the sequential loop, which reproduces the reference's behaviour entry by entry.
bool parse_entries_parallel(const char *p, const char *end, long NZ, Field field, long M, long N, bool mirrored,
                            int32_t *ri, int32_t *ci, double *v)
{
    const size_t bytes = (size_t)(end - p);
    unsigned nt = std::thread::hardware_concurrency();
    if (const char *e = getenv("SBLAS_LOADER_THREADS")) nt = (unsigned)atoi(e);
    if (nt > 32) nt = 32;
    size_t min_bytes = 64u << 20; // below this the sequential loop is as fast as starting threads
    if (const char *e = getenv("SBLAS_LOADER_MIN_BYTES")) min_bytes = (size_t)strtoull(e, nullptr, 10); // (tests)
    if (nt < 2 || bytes < min_bytes || NZ < (long)nt) return false;
    const int tpe = field == F_PATTERN ? 2 : field == F_COMPLEX ? 4 : 3;
    // chunk starts, moved forward to the first byte after a whitespace character (so that no token is cut)
    std::vector<const char *> cb(nt + 1);
    cb[0] = p;
    cb[nt] = end;
    for (unsigned t = 1; t < nt; ++t) {
        const char *q = p + bytes / nt * t;
        while (q < end && !isspace((unsigned char)*q)) ++q;
        cb[t] = q;
    }
    std::vector<long long> ntok(nt, 0);
    auto count = [&](unsigned t) {
        long long c = 0;
        bool in = false;
        for (const char *q = cb[t]; q < cb[t + 1]; ++q) {
            const bool ws = isspace((unsigned char)*q);
            c += (!ws && !in);
            in = !ws;
        }
        ntok[t] = c;
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(count, t);
        for (auto &x : th) x.join();
    }
    std::vector<long long> first(nt + 1, 0); // tokens in front of chunk t
    for (unsigned t = 0; t < nt; ++t) first[t + 1] = first[t] + ntok[t];
    if (first[nt] < (long long)NZ * tpe) return false;
    std::atomic<bool> bad{false};
    auto convert = [&](unsigned t) {
        // first entry that STARTS in this chunk, and one past the last
        long long e0 = (first[t] + tpe - 1) / tpe, e1 = (first[t + 1] + tpe - 1) / tpe;
        if (e1 > NZ) e1 = NZ;
        if (e0 >= e1) return;
        const char *q = cb[t];
        for (long long skip = e0 * tpe - first[t]; skip > 0; --skip) { // tokens of an entry the chunk before finishes
            q = skip_ws(q, end);
            q = token_end(q, end);
        }
        for (long long e = e0; e < e1; ++e) {
            if (!parse_entry<true>(q, end, field, M, N, mirrored, ri[(size_t)e], ci[(size_t)e], v[(size_t)e])) {
                bad.store(true);
                return;
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(convert, t);
        for (auto &x : th) x.join();
    }
    return !bad.load();
}

// SBLAS_LOADER_TIMING=1: phase times of a parse on stderr (tools/loader_bench.py)
struct PhaseClock {
    bool on;
    std::chrono::steady_clock::time_point t;
    PhaseClock() : on(getenv("SBLAS_LOADER_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
    void lap(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  loader: %-28s %.2f s\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

int parse_file(const char *path, Parsed &out)
{
    PhaseClock clock;
    FILE *f = fopen(path, "rb");
    if (!f) return SBLAS_E_IO;
    struct stat st;
    if (fstat(fileno(f), &st) != 0) { fclose(f); return SBLAS_E_IO; }
    // (plain arrays, not std::vector: a vector would zero-fill -- and page in -- a gigabyte that fread overwrites)
    std::unique_ptr<char[]> buf(new char[(size_t)st.st_size + 1]);
    const size_t got = fread(buf.get(), 1, (size_t)st.st_size, f);
    fclose(f);
    buf[got] = '\0';
    const char *p = buf.get(), *end = buf.get() + got;
    clock.lap("read file");

    // banner (mmio.h:254-337): five tokens on the first line, the last four case-insensitive
    const char *eol = (const char *)memchr(p, '\n', (size_t)(end - p));
    const char *line_end = eol ? eol : end;
    std::string tok[5];
    {
        const char *q = p;
        for (int i = 0; i < 5; ++i) {
            q = skip_ws(q, line_end);
            const char *e = token_end(q, line_end);
            if (e == q) return SBLAS_E_IO;
            tok[i] = (i == 0) ? std::string(q, e) : lowered(q, e);
            q = e;
        }
    }
    if (tok[0].compare(0, 14, "%%MatrixMarket") != 0) return SBLAS_E_IO;
    if (tok[1] != "matrix") return SBLAS_E_IO;
    if (tok[2] != "coordinate" && tok[2] != "array") return SBLAS_E_IO;
    Field field;
    if (tok[3] == "real") field = F_REAL;
    else if (tok[3] == "complex") field = F_COMPLEX;
    else if (tok[3] == "pattern") field = F_PATTERN;
    else if (tok[3] == "integer") field = F_INTEGER;
    else return SBLAS_E_IO;
    bool mirrored;
    if (tok[4] == "general" || tok[4] == "skew-symmetric") mirrored = false; // skew is NOT mirrored (:46-51)
    else if (tok[4] == "symmetric" || tok[4] == "hermitian") mirrored = true;
    else return SBLAS_E_IO;
    p = eol ? eol + 1 : end;

    // size line (mmio.h:339-367): skip '%' lines; first other line; else keep scanning ints
    long M = 0, N = 0, NZ = 0;
    for (;;) {
        if (p >= end) return SBLAS_E_IO;
        eol = (const char *)memchr(p, '\n', (size_t)(end - p));
        line_end = eol ? eol : end;
        const bool comment = (*p == '%');
        const char *q = p;
        p = eol ? eol + 1 : end;
        if (comment) continue;
        const char *t = q;
        if (parse_int(t, line_end, M) && parse_int(t, line_end, N) && parse_int(t, line_end, NZ)) break;
        // blank / odd line: the reference falls back to fscanf("%d %d %d") on the stream
        if (!(parse_int(p, end, M) && parse_int(p, end, N) && parse_int(p, end, NZ))) return SBLAS_E_IO;
        break;
    }
    if (M < 0 || N < 0 || NZ < 0 || M > 0x7ffffffe || N > 0x7fffffff || NZ > 0x7fffffff) return SBLAS_E_IO;

    // (uninitialised: the tokeniser threads touch their own parts first)
    std::unique_ptr<int32_t[]> ri(new int32_t[(size_t)NZ + 1]), ci(new int32_t[(size_t)NZ + 1]);
    std::unique_ptr<double[]> v(new double[(size_t)NZ + 1]);
    std::vector<int32_t> rowptr((size_t)M + 1, 0);
    if (!parse_entries_parallel(p, end, NZ, field, M, N, mirrored, ri.get(), ci.get(), v.get())) {
        for (long e = 0; e < NZ; ++e)
            if (!parse_entry(p, end, field, M, N, mirrored, ri[(size_t)e], ci[(size_t)e], v[(size_t)e])) return SBLAS_E_IO;
    }
    clock.lap("tokenise entries");
    for (long e = 0; e < NZ; ++e) rowptr[(size_t)ri[(size_t)e]]++;
    if (mirrored)
        for (long e = 0; e < NZ; ++e)
            if (ri[(size_t)e] != ci[(size_t)e]) rowptr[(size_t)ci[(size_t)e]]++;
    // exclusive scan
    long long run = 0;
    for (long r = 0; r <= M; ++r) {
        const long long here = rowptr[(size_t)r];
        if (run > 0x7fffffffLL) return SBLAS_E_IO;
        rowptr[(size_t)r] = (int32_t)run;
        run += here;
    }
    const int32_t nnz = rowptr[(size_t)M];
    out.colidx.assign((size_t)nnz, 0);
    out.val.assign((size_t)nnz, 0.0);
    std::vector<int32_t> cursor(rowptr.begin(), rowptr.end() - 1);
    for (long e = 0; e < NZ; ++e) {
        const int32_t i = ri[(size_t)e], j = ci[(size_t)e];
        int32_t at = cursor[(size_t)i]++;
        out.colidx[(size_t)at] = j;
        out.val[(size_t)at] = v[(size_t)e];
        if (mirrored && i != j) { // the (j,i) twin follows its original immediately (:242-262)
            at = cursor[(size_t)j]++;
            out.colidx[(size_t)at] = i;
            out.val[(size_t)at] = v[(size_t)e];
        }
    }
    clock.lap("count, scan, scatter");
    out.rowptr.swap(rowptr);
    out.rows = (int32_t)M;
    out.cols = (int32_t)N;
    out.nnz = nnz;
    out.mirrored = mirrored ? 1 : 0;
    out.path = path;
    out.size = (long long)st.st_size;
    out.mtime_ns = (long long)st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec;
    return SBLAS_OK;
}

// ------------------------------------------------------------------------------------------
// Binary sidecar cache: with SBLAS_CSR_CACHE=1 a parsed matrix is written next to its source as
// <path>.csrbin and read back on later loads when the source's size and mtime still match -- the text parse of a
// Queen_4147-sized file takes minutes, the binary read seconds.  The arrays are the loader's own output, so the
// result is bit-identical by construction (tests compare both ways).  Opt-in: it writes a file beside the input.
// Layout: 8 x int64 header {magic, version, source size, source mtime_ns, rows, cols, nnz, mirrored},
// rowptr[rows+1] int32, colidx[nnz] int32, (padding to 8 bytes,) val[nnz] fp64.
// ------------------------------------------------------------------------------------------
constexpr long long CSRBIN_MAGIC = 0x4e4942525343424cLL; // "LBCSRBIN" little endian
constexpr long long CSRBIN_VERSION = 1;

bool cache_enabled()
{
    const char *e = getenv("SBLAS_CSR_CACHE");
    return e && *e && strcmp(e, "0") != 0;
}

bool read_sidecar(const std::string &bin, long long size, long long mtime_ns, Parsed &out)
{
    FILE *f = fopen(bin.c_str(), "rb");
    if (!f) return false;
    long long h[8];
    bool ok = fread(h, sizeof(long long), 8, f) == 8 && h[0] == CSRBIN_MAGIC && h[1] == CSRBIN_VERSION && h[2] == size &&
              h[3] == mtime_ns && h[4] >= 0 && h[5] >= 0 && h[6] >= 0 && h[4] < 0x7fffffffLL && h[6] <= 0x7fffffffLL;
    if (ok) {
        out.rows = (int32_t)h[4];
        out.cols = (int32_t)h[5];
        out.nnz = (int32_t)h[6];
        out.mirrored = (int32_t)h[7];
        out.rowptr.resize((size_t)out.rows + 1);
        out.colidx.resize((size_t)out.nnz);
        out.val.resize((size_t)out.nnz);
        const size_t ints = (size_t)out.rows + 1 + (size_t)out.nnz;
        ok = fread(out.rowptr.data(), 4, out.rowptr.size(), f) == out.rowptr.size() &&
             fread(out.colidx.data(), 4, out.colidx.size(), f) == out.colidx.size();
        int32_t padw = 0;
        if (ok && (ints & 1)) ok = fread(&padw, 4, 1, f) == 1;
        ok = ok && fread(out.val.data(), 8, out.val.size(), f) == out.val.size();
        // structural sanity: a truncated or foreign file must not reach the caller
        ok = ok && out.rowptr[0] == 0 && out.rowptr[out.rows] == out.nnz;
    }
    fclose(f);
    return ok;
}

void write_sidecar(const std::string &bin, const Parsed &p)
{
    const std::string tmp = bin + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return; // read-only directory: the cache is an optimisation, not a requirement
    const long long h[8] = {CSRBIN_MAGIC, CSRBIN_VERSION, p.size, p.mtime_ns, p.rows, p.cols, p.nnz, p.mirrored};
    const size_t ints = (size_t)p.rows + 1 + (size_t)p.nnz;
    const int32_t padw = 0;
    bool ok = fwrite(h, sizeof(long long), 8, f) == 8 && fwrite(p.rowptr.data(), 4, p.rowptr.size(), f) == p.rowptr.size() &&
              fwrite(p.colidx.data(), 4, p.colidx.size(), f) == p.colidx.size();
    if (ok && (ints & 1)) ok = fwrite(&padw, 4, 1, f) == 1;
    ok = ok && fwrite(p.val.data(), 8, p.val.size(), f) == p.val.size();
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tmp.c_str(), bin.c_str()) == 0;
    if (!ok) remove(tmp.c_str());
}

// returns with g_cache_mu held by the caller
int ensure_parsed(const char *path)
{
    struct stat st;
    if (!path || stat(path, &st) != 0) return SBLAS_E_IO;
    const long long mt = (long long)st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec;
    if (g_cache.size == (long long)st.st_size && g_cache.mtime_ns == mt && g_cache.path == path) return SBLAS_OK;
    Parsed fresh;
    const bool use_cache = cache_enabled();
    const std::string bin = std::string(path) + ".csrbin";
    if (use_cache && read_sidecar(bin, (long long)st.st_size, mt, fresh)) {
        fresh.path = path;
        fresh.size = (long long)st.st_size;
        fresh.mtime_ns = mt;
        g_cache = std::move(fresh);
        return SBLAS_OK;
    }
    fresh = Parsed();
    const int rc = parse_file(path, fresh);
    if (rc != SBLAS_OK) return rc;
    if (use_cache) write_sidecar(bin, fresh);
    g_cache = std::move(fresh);
    return SBLAS_OK;
}

} // namespace

extern "C" int sblas_mm_read_info(const char *path, int32_t *rows, int32_t *cols, int32_t *nnz,
                                  int32_t *is_symmetric)
{
    std::lock_guard<std::mutex> lock(g_cache_mu);
    const int rc = ensure_parsed(path);
    if (rc != SBLAS_OK) return rc;
    if (rows) *rows = g_cache.rows;
    if (cols) *cols = g_cache.cols;
    if (nnz) *nnz = g_cache.nnz;
    if (is_symmetric) *is_symmetric = g_cache.mirrored;
    return SBLAS_OK;
}

extern "C" int sblas_mm_read_csr(const char *path, int32_t *rowptr, int32_t *colidx, double *val)
{
    std::lock_guard<std::mutex> lock(g_cache_mu);
    const int rc = ensure_parsed(path);
    if (rc != SBLAS_OK) return rc;
    if (!rowptr) return SBLAS_E_INVALID;
    memcpy(rowptr, g_cache.rowptr.data(), g_cache.rowptr.size() * sizeof(int32_t));
    if (g_cache.nnz > 0) {
        if (!colidx || !val) return SBLAS_E_INVALID;
        memcpy(colidx, g_cache.colidx.data(), (size_t)g_cache.nnz * sizeof(int32_t));
        memcpy(val, g_cache.val.data(), (size_t)g_cache.nnz * sizeof(double));
    }
    // the parse is handed over once; drop it so a large matrix is not held twice
    g_cache = Parsed();
    return SBLAS_OK;
}
