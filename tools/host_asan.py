"""Host code under ASan + UBSan (CPU only; GPU sanitizers are not available on the pool): the loader (sequential and threaded
tokenizer, sidecar cache, malformed files), the partitioners and the initialiser, through ctypes on a sanitizer build of
host_util.cpp.

  g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -o /tmp/libhost_asan.so \
      s-blas_amd/csrc/host_util.cpp -lpthread
  LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
      python tools/host_asan.py /tmp/libhost_asan.so"""
import ctypes as C, numpy as np, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else "/tmp/libhost_asan.so")
L.sblas_mm_read_info.argtypes = [C.c_char_p] + [C.POINTER(C.c_int32)] * 4
L.sblas_mm_read_csr.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
L.sblas_partition_nnz.restype = C.c_int64
L.sblas_partition_nnz.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int, C.c_int] + [C.c_void_p] * 5
L.sblas_find_row_of_nnz.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
L.sblas_partition_dense.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
L.sblas_host_fill_rand0to1.argtypes = [C.c_void_p, C.c_int64, C.c_uint]
def read(path):
    r, c, z, s = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    rc = L.sblas_mm_read_info(os.fsencode(path), C.byref(r), C.byref(c), C.byref(z), C.byref(s))
    if rc: return rc, None
    rp = np.zeros(r.value + 1, np.int32); ci = np.zeros(max(z.value, 1), np.int32); v = np.zeros(max(z.value, 1))
    rc = L.sblas_mm_read_csr(os.fsencode(path), rp.ctypes.data, ci.ctypes.data, v.ctypes.data)
    return rc, (r.value, c.value, z.value, s.value, rp, ci[:z.value], v[:z.value])
rc, m = read(os.path.join(ROOT, "tests", "golden", "ash85.mtx")); assert rc == 0 and m[2] == 523, (rc, m and m[:4])
rng = np.random.default_rng(1)
tmp = tempfile.mkdtemp()
def write(path, rows, cols, ents, field="real", sym="general", glue=False, crlf=False, comments=True):
    nl = "\r\n" if crlf else "\n"
    with open(path, "w", newline="") as f:
        f.write("%%MatrixMarket matrix coordinate " + field + " " + sym + nl)
        if comments: f.write("% a comment" + nl + "%" + nl)
        f.write("%d %d %d%s" % (rows, cols, len(ents), nl))
        for (i, j, x) in ents:
            if field == "pattern": f.write("%d %d%s" % (i, j, nl))
            elif field == "integer": f.write("%d %d %d%s" % (i, j, int(x * 10), nl))
            elif glue: f.write("%d\t%d   %.17g%s" % (i, j, x, nl))
            else: f.write("%d %d %.17g%s" % (i, j, x, nl))
n_ok = 0
for case in range(60):
    rows, cols = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    k = int(rng.integers(0, 3000))
    sym = ["general", "symmetric", "skew-symmetric"][case % 3] if rows == cols else "general"
    ents = set()
    while len(ents) < min(k, rows * cols // 2):
        i, j = int(rng.integers(1, rows + 1)), int(rng.integers(1, cols + 1))
        if sym != "general" and j > i: i, j = j, i
        if sym == "skew-symmetric" and i == j: continue
        ents.add((i, j))
    ents = [(i, j, float(rng.standard_normal())) for (i, j) in ents]
    p = os.path.join(tmp, "c%d.mtx" % case)
    write(p, rows, cols, ents, field=["real", "pattern", "integer"][case % 3 if case % 5 else 0], sym=sym, glue=case % 4 == 1, crlf=case % 7 == 3)
    for thr in ("1", "4"):
        os.environ["SBLAS_LOADER_THREADS"] = thr
        os.environ["SBLAS_LOADER_MIN_BYTES"] = "0"
        rc, m = read(p)
        assert rc == 0, (case, thr, rc)
        assert m[4][-1] == m[2] and (np.diff(m[4]) >= 0).all(), case
    n_ok += 1
# sidecar cache: second read comes from the cache file
os.environ["SBLAS_CSR_CACHE"] = "1"
for _ in range(2):
    rc, m2 = read(os.path.join(tmp, "c5.mtx")); assert rc == 0
os.environ.pop("SBLAS_CSR_CACHE")
# malformed files: error codes, no crashes
bad = {"empty": "", "banner": "%%MatrixMarket matrix array real general\n2 2\n1\n", "short": "%%MatrixMarket matrix coordinate real general\n3 3 5\n1 1 1.0\n",
       "range": "%%MatrixMarket matrix coordinate real general\n3 3 1\n4 1 1.0\n", "junk": "%%MatrixMarket matrix coordinate real general\n3 3 1\n1 x 1.0\n",
       "neg": "%%MatrixMarket matrix coordinate real general\n-3 3 1\n1 1 1.0\n", "huge": "%%MatrixMarket matrix coordinate real general\n3 3 99999999999\n1 1 1.0\n",
       "nosize": "%%MatrixMarket matrix coordinate real general\n% only comments\n", "trunc": "%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1.0\n2 2"}
for name, text in bad.items():
    p = os.path.join(tmp, "bad_%s.mtx" % name); open(p, "w").write(text)
    for thr in ("1", "4"):
        os.environ["SBLAS_LOADER_THREADS"] = thr
        rc, m = read(p)
        assert rc != 0, (name, thr)
rc, m = read(os.path.join(tmp, "does_not_exist.mtx")); assert rc != 0
# partitioners
for case in range(200):
    rows = int(rng.integers(1, 300)); lens = rng.integers(0, 9, rows); lens[rng.integers(0, rows)] += int(rng.integers(0, 500))
    rp = np.zeros(rows + 1, np.int32); np.cumsum(lens, out=rp[1:]); nnz = int(rp[-1])
    for g in (1, 2, 3, 8):
        tot = 0
        for i in range(g):
            s, e, k = C.c_int32(), C.c_int32(), C.c_int32(); f = C.c_int64(); buf = np.zeros(rows + 2, np.int32)
            num = L.sblas_partition_nnz(rp.ctypes.data, rows, nnz, g, i, C.byref(s), C.byref(e), C.byref(k), C.byref(f), buf.ctypes.data)
            assert num >= 0; tot += k.value
        assert tot == nnz, (case, g, tot, nnz)
    if nnz:
        for q in rng.integers(0, nnz, 20):
            r = L.sblas_find_row_of_nnz(rp.ctypes.data, rows, int(q)); assert rp[r] <= q < rp[r + 1]
for fo in (0, 1, 7, 64, 65, 1000003):
    for g in (1, 2, 3, 8):
        tot = 0
        for i in range(g):
            o, d = C.c_int64(), C.c_int64(); assert L.sblas_partition_dense(fo, g, i, C.byref(o), C.byref(d)) == 0; tot += d.value; assert d.value >= 0
        assert tot == fo
out = np.empty(1000); assert L.sblas_host_fill_rand0to1(out.ctypes.data, 1000, 211) == 0
print("host code under ASan + UBSan: %d loader cases x 2 tokenizers, %d malformed files, partitioners, initialiser: clean" % (n_ok, len(bad)))
