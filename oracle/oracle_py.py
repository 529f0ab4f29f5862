"""TEST INFRASTRUCTURE ONLY: ctypes view of oracle/liboracle.so (our CPU restatement of the
reference) and, when built, oracle/_ref/libref_loader.so (the reference's own loader).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "sblas_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"], stdout=subprocess.DEVNULL)
    return so


def _load():
    lib = C.CDLL(build())
    lib.orc_mm_info.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 4
    lib.orc_mm_info.restype = C.c_int
    lib.orc_mm_data.argtypes = [C.c_char_p, _i32p, _i32p, _f64p]
    lib.orc_mm_data.restype = C.c_int
    lib.orc_spmm_csr.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_double, C.c_double]
    lib.orc_spmm_csr.restype = None
    lib.orc_spmm_csr_omp.argtypes = lib.orc_spmm_csr.argtypes
    lib.orc_spmm_csr_omp.restype = None
    lib.orc_spmm_csr_rows.argtypes = [C.c_int, C.c_int] + lib.orc_spmm_csr.argtypes
    lib.orc_spmm_csr_rows.restype = None
    lib.orc_spmv_csr.argtypes = [C.c_int, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_double, C.c_double]
    lib.orc_spmv_csr.restype = None
    lib.orc_axpby.argtypes = [C.c_size_t, C.c_double, _f64p, C.c_double, _f64p]
    lib.orc_axpby.restype = None
    lib.orc_fill_rand0to1.argtypes = [_f64p, C.c_size_t]
    lib.orc_fill_rand0to1.restype = None
    lib.orc_check_equal.argtypes = [_f64p, _f64p, C.c_size_t]
    lib.orc_check_equal.restype = C.c_int
    lib.orc_find_row.argtypes = [_i32p, C.c_int, C.c_int]
    lib.orc_find_row.restype = C.c_int
    lib.orc_avg_nnz_float.argtypes = [C.c_int, C.c_int]
    lib.orc_avg_nnz_float.restype = C.c_int
    lib.orc_avg_nnz_exact.argtypes = [C.c_int, C.c_int]
    lib.orc_avg_nnz_exact.restype = C.c_int
    lib.orc_partition_nnz.argtypes = [_i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
    lib.orc_partition_nnz.restype = C.c_int
    lib.orc_partition_dense.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.orc_partition_dense.restype = None
    lib.orc_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
    lib.orc_fnv1a64.restype = C.c_uint64
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def ref_loader():
    """The reference's own mmio loader (oracle/_ref), or None when it was never built."""
    so = os.path.join(_HERE, "_ref", "libref_loader.so")
    if not os.path.exists(so):
        return None
    r = C.CDLL(so)
    r.ref_mm_info.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 4
    r.ref_mm_info.restype = C.c_int
    r.ref_mm_data.argtypes = [C.c_char_p, _i32p, _i32p, _f64p]
    r.ref_mm_data.restype = C.c_int
    return r


def _read_with(info_fn, data_fn, path):
    m, n, nnz, sym = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = info_fn(path.encode(), C.byref(m), C.byref(n), C.byref(nnz), C.byref(sym))
    if rc != 0:
        raise IOError("mm_info(%s) -> %d" % (path, rc))
    rowptr = np.zeros(m.value + 1, np.int32)
    colidx = np.zeros(max(nnz.value, 1), np.int32)
    val = np.zeros(max(nnz.value, 1), np.float64)
    rc = data_fn(path.encode(), rowptr, colidx, val)
    if rc != 0:
        raise IOError("mm_data(%s) -> %d" % (path, rc))
    return m.value, n.value, nnz.value, sym.value, rowptr, colidx[:nnz.value].copy(), val[:nnz.value].copy()


def read_mtx(path):
    L = lib()
    return _read_with(L.orc_mm_info, L.orc_mm_data, path)


def read_mtx_ref(path):
    r = ref_loader()
    return _read_with(r.ref_mm_info, r.ref_mm_data, path)


def spmm(M, K, N, rowptr, colidx, val, B, C_, alpha, beta):
    """In place on C_ (column-major flat array of M*N)."""
    lib().orc_spmm_csr(M, K, N, rowptr, colidx, val, B, C_, alpha, beta)
    return C_


def spmm_omp(M, K, N, rowptr, colidx, val, B, C_, alpha, beta):
    lib().orc_spmm_csr_omp(M, K, N, rowptr, colidx, val, B, C_, alpha, beta)
    return C_


def spmm_rows(r0, r1, M, K, N, rowptr, colidx, val, B, C_, alpha, beta):
    lib().orc_spmm_csr_rows(r0, r1, M, K, N, rowptr, colidx, val, B, C_, alpha, beta)
    return C_


def spmv(M, rowptr, colidx, val, x, y, alpha, beta):
    lib().orc_spmv_csr(M, rowptr, colidx, val, x, y, alpha, beta)
    return y


def rand0to1(n):
    v = np.empty(n, np.float64)
    lib().orc_fill_rand0to1(v, n)
    return v


def fnv(arr):
    a = np.ascontiguousarray(arr)
    return "%016x" % lib().orc_fnv1a64(a.ctypes.data, a.nbytes)


def partition_nnz(rowptr, M, nnz, g, i, exact=False):
    L = lib()
    avg = L.orc_avg_nnz_exact(nnz, g) if exact else L.orc_avg_nnz_float(nnz, g)
    s, e, k = C.c_int(), C.c_int(), C.c_int()
    buf = np.zeros(M + 2, np.int32)
    num = L.orc_partition_nnz(rowptr, M, nnz, g, i, avg, C.byref(s), C.byref(e), C.byref(k), buf.ctypes.data)
    if num < 0:
        raise ValueError("partition failed")
    return s.value, e.value, k.value, buf[:num].copy(), avg


def partition_dense(first, g, i):
    o, d = C.c_int(), C.c_int()
    lib().orc_partition_dense(first, g, i, C.byref(o), C.byref(d))
    return o.value, d.value


# ---- the other instantiations of the reference's templates (float values and / or 64-bit indices) ----
_TYPED = {(np.dtype(np.float32), np.dtype(np.int32)): "f32_i32", (np.dtype(np.float64), np.dtype(np.int64)): "f64_i64",
          (np.dtype(np.float32), np.dtype(np.int64)): "f32_i64"}


def _typed_fn(kind, val, rowptr):
    name = _TYPED[(val.dtype, rowptr.dtype)]
    fn = getattr(lib(), "orc_%s_csr_%s" % (kind, name))
    ct = C.c_float if val.dtype == np.float32 else C.c_double
    ip = np.ctypeslib.ndpointer(rowptr.dtype, flags="C_CONTIGUOUS")
    vp = np.ctypeslib.ndpointer(val.dtype, flags="C_CONTIGUOUS")
    fn.restype = None
    if kind == "spmm":
        fn.argtypes = [C.c_int64, C.c_int64, C.c_int64, ip, ip, vp, vp, vp, ct, ct]
    else:
        fn.argtypes = [C.c_int64, ip, ip, vp, vp, vp, ct, ct]
    return fn


def spmm_typed(M, K, N, rowptr, colidx, val, B, C_, alpha, beta):
    """sblas_spmm_csr_cpu<IdxType, DataType> for (float32 | float64) x (int32 | int64) arrays; in place on C_."""
    if val.dtype == np.float64 and rowptr.dtype == np.int32:
        return spmm(M, K, N, rowptr, colidx, val, B, C_, alpha, beta)
    _typed_fn("spmm", val, rowptr)(M, K, N, rowptr, colidx, val, B, C_, alpha, beta)
    return C_


def spmv_typed(M, rowptr, colidx, val, x, y, alpha, beta):
    if val.dtype == np.float64 and rowptr.dtype == np.int32:
        return spmv(M, rowptr, colidx, val, x, y, alpha, beta)
    _typed_fn("spmv", val, rowptr)(M, rowptr, colidx, val, x, y, alpha, beta)
    return y
