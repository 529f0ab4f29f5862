"""ctypes binding of libsblas_hip.so (the C ABI in include/sblas_hip.h).

Plumbing only: torch supplies device memory and streams, every compute call goes through the
shared library.  There is no CPU or torch fallback here -- if the HIP library is missing, or a
tensor is not on the GPU, the call raises.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.normpath(os.path.join(_PKG, "..", ".."))
# SBLAS_LIB_PATH: A/B runs of two builds of the library (experiments only)
LIB_PATH = os.environ.get("SBLAS_LIB_PATH") or os.path.join(_ROOT, "lib", "libsblas_hip.so")

# every symbol include/sblas_hip.h declares (tests check that the .so exports all of them)
EXPORTS = [
    "sblas_hip_version", "sblas_hip_error_string", "sblas_hip_device_count",
    "sblas_hip_spmm_csr_f64_i32_workspace", "sblas_hip_spmm_csr_f64_i32", "sblas_hip_spmm_ldbt",
    "sblas_hip_dense_to_rowmajor_f64", "sblas_hip_spmm_csr_rowmajorB_f64_i32",
    "sblas_hip_debug_spmm_panel_stats", "sblas_hip_debug_reload_env",
    "sblas_hip_debug_spmm_kernel_events", "sblas_hip_debug_spmm_last_kernel_ms", "sblas_hip_spmv_csr_f64_i32", "sblas_hip_axpby_f64",
    "sblas_hip_comm_get", "sblas_hip_comm_release_all", "sblas_hip_allreduce_sum_f64",
    "sblas_hip_merge_rowblocks_f64", "sblas_hip_merge_rowblocks_local_f64",
    "sblas_find_row_of_nnz", "sblas_partition_nnz", "sblas_partition_dense",
    "sblas_mm_read_info", "sblas_mm_read_csr", "sblas_host_fill_rand0to1",
    "sblas_hip_spmm_csr_workspace", "sblas_hip_spmm_csr", "sblas_hip_spmv_csr", "sblas_hip_axpby",
    "sblas_hip_allreduce_sum", "sblas_hip_merge_rowblocks", "sblas_partition_nnz_i64",
    "sblas_hip_debug_validate_csr_i32",
    "sblas_hip_spmm_plan_create", "sblas_hip_spmm_plan_destroy", "sblas_hip_spmm_plan_info", "sblas_hip_spmm_csr_f64_i32_planned",
]


class SblasError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SblasError("libsblas_hip.so not built (%s): run __graft_entry__.build() / make -C s-blas_amd" % LIB_PATH)
    # torch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1 / librccl.so.1.  Load torch FIRST so
    # that our DT_NEEDED entries (and the dlopen of RCCL) resolve by SONAME to the copies already in the process:
    # two HIP runtimes in one process do not share a device context.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    i64, i32, f64, vp, sz = C.c_int64, C.c_int32, C.c_double, C.c_void_p, C.c_size_t
    L.sblas_hip_version.restype = C.c_int
    L.sblas_hip_error_string.restype = C.c_char_p
    L.sblas_hip_error_string.argtypes = [C.c_int]
    L.sblas_hip_device_count.restype = C.c_int
    L.sblas_hip_spmm_ldbt.restype = i64
    L.sblas_hip_spmm_ldbt.argtypes = [i64]
    L.sblas_hip_spmm_csr_f64_i32_workspace.restype = sz
    L.sblas_hip_spmm_csr_f64_i32_workspace.argtypes = [i64, i64, i64, i64]
    L.sblas_hip_spmm_csr_f64_i32.restype = C.c_int
    L.sblas_hip_spmm_csr_f64_i32.argtypes = [C.c_int, vp, i64, i64, i64, vp, vp, vp, vp, i64, i64, f64, f64, vp, i64, vp, sz]
    L.sblas_hip_debug_validate_csr_i32.restype = C.c_int
    L.sblas_hip_debug_validate_csr_i32.argtypes = [C.c_int, vp, i64, i64, i64, vp, vp]
    L.sblas_hip_spmm_plan_create.restype = C.c_int
    L.sblas_hip_spmm_plan_create.argtypes = [C.c_int, vp, i64, i64, i64, vp, vp, i64, C.POINTER(vp)]
    L.sblas_hip_spmm_plan_destroy.restype = C.c_int
    L.sblas_hip_spmm_plan_destroy.argtypes = [vp]
    L.sblas_hip_spmm_plan_info.restype = C.c_int
    L.sblas_hip_spmm_plan_info.argtypes = [vp, C.POINTER(i64)]
    L.sblas_hip_spmm_csr_f64_i32_planned.restype = C.c_int
    L.sblas_hip_spmm_csr_f64_i32_planned.argtypes = [vp, C.c_int, vp, i64, i64, i64, vp, vp, vp, vp, i64, i64, f64, f64, vp, i64, vp, sz]
    L.sblas_hip_dense_to_rowmajor_f64.restype = C.c_int
    L.sblas_hip_dense_to_rowmajor_f64.argtypes = [C.c_int, vp, i64, i64, vp, i64, vp, i64]
    L.sblas_hip_spmm_csr_rowmajorB_f64_i32.restype = C.c_int
    L.sblas_hip_spmm_csr_rowmajorB_f64_i32.argtypes = [C.c_int, vp, i64, i64, i64, vp, vp, vp, vp, i64, i64, f64, f64, vp, i64]
    L.sblas_hip_debug_reload_env.restype = C.c_int
    L.sblas_hip_debug_reload_env.argtypes = []
    L.sblas_hip_debug_spmm_panel_stats.restype = C.c_int
    L.sblas_hip_debug_spmm_panel_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    L.sblas_hip_debug_spmm_kernel_events.restype = C.c_int
    L.sblas_hip_debug_spmm_kernel_events.argtypes = [C.c_int]
    L.sblas_hip_debug_spmm_last_kernel_ms.restype = C.c_int
    L.sblas_hip_debug_spmm_last_kernel_ms.argtypes = [C.POINTER(C.c_float)]
    L.sblas_hip_spmv_csr_f64_i32.restype = C.c_int
    L.sblas_hip_spmv_csr_f64_i32.argtypes = [C.c_int, vp, i64, i64, i64, vp, vp, vp, vp, f64, f64, vp]
    L.sblas_hip_axpby_f64.restype = C.c_int
    L.sblas_hip_axpby_f64.argtypes = [C.c_int, vp, i64, f64, vp, f64, vp]
    L.sblas_hip_comm_get.restype = C.c_int
    L.sblas_hip_comm_get.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    L.sblas_hip_comm_release_all.restype = None
    L.sblas_hip_allreduce_sum_f64.restype = C.c_int
    L.sblas_hip_allreduce_sum_f64.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), i64]
    L.sblas_hip_merge_rowblocks_f64.restype = C.c_int
    L.sblas_hip_merge_rowblocks_f64.argtypes = [vp, i64, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(vp), C.POINTER(vp),
                                                f64, f64, C.POINTER(vp), i64, C.POINTER(vp)]
    L.sblas_hip_merge_rowblocks_local_f64.restype = C.c_int
    L.sblas_hip_merge_rowblocks_local_f64.argtypes = [C.c_int, vp, i64, i64, C.c_int, C.POINTER(i64), C.POINTER(i64),
                                                      C.POINTER(vp), f64, f64, vp, i64]
    L.sblas_find_row_of_nnz.restype = i32
    L.sblas_find_row_of_nnz.argtypes = [vp, i32, i32]
    L.sblas_partition_nnz.restype = i64
    L.sblas_partition_nnz.argtypes = [vp, i32, i32, C.c_int, C.c_int, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i64), vp]
    L.sblas_partition_dense.restype = C.c_int
    L.sblas_partition_dense.argtypes = [i64, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]
    L.sblas_mm_read_info.restype = C.c_int
    L.sblas_mm_read_info.argtypes = [C.c_char_p] + [C.POINTER(i32)] * 4
    L.sblas_mm_read_csr.restype = C.c_int
    L.sblas_mm_read_csr.argtypes = [C.c_char_p, vp, vp, vp]
    L.sblas_host_fill_rand0to1.restype = C.c_int
    L.sblas_host_fill_rand0to1.argtypes = [vp, i64, C.c_uint]
    L.sblas_hip_spmm_csr_workspace.restype = sz
    L.sblas_hip_spmm_csr_workspace.argtypes = [C.c_int, C.c_int, i64, i64, i64, i64]
    L.sblas_hip_spmm_csr.restype = C.c_int
    L.sblas_hip_spmm_csr.argtypes = [C.c_int, vp, C.c_int, C.c_int, i64, i64, i64, vp, vp, vp, vp, i64, i64, f64, f64, vp, i64, vp, sz]
    L.sblas_hip_spmv_csr.restype = C.c_int
    L.sblas_hip_spmv_csr.argtypes = [C.c_int, vp, C.c_int, C.c_int, i64, i64, i64, vp, vp, vp, vp, f64, f64, vp]
    L.sblas_hip_axpby.restype = C.c_int
    L.sblas_hip_axpby.argtypes = [C.c_int, vp, C.c_int, i64, f64, vp, f64, vp]
    L.sblas_hip_allreduce_sum.restype = C.c_int
    L.sblas_hip_allreduce_sum.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), i64]
    L.sblas_hip_merge_rowblocks.restype = C.c_int
    L.sblas_hip_merge_rowblocks.argtypes = [vp, C.c_int, i64, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(vp), C.POINTER(vp),
                                            f64, f64, C.POINTER(vp), i64, C.POINTER(vp)]
    L.sblas_partition_nnz_i64.restype = i64
    L.sblas_partition_nnz_i64.argtypes = [vp, i64, i64, C.c_int, C.c_int] + [C.POINTER(i64)] * 4 + [vp]
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise SblasError("%s failed: %s (code %d)" % (what, lib().sblas_hip_error_string(rc).decode(), rc))


# ------------------------------------------------------------------------------------------
# host-side pure functions
# ------------------------------------------------------------------------------------------
def read_mtx(path):
    """MatrixMarket -> (rows, cols, nnz, symmetric, rowptr[int32], colidx[int32], val[float64])."""
    L = lib()
    r, c, z, s = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    check(L.sblas_mm_read_info(os.fsencode(path), C.byref(r), C.byref(c), C.byref(z), C.byref(s)), "sblas_mm_read_info")
    rowptr = np.zeros(r.value + 1, np.int32)
    colidx = np.zeros(max(z.value, 1), np.int32)
    val = np.zeros(max(z.value, 1), np.float64)
    check(L.sblas_mm_read_csr(os.fsencode(path), rowptr.ctypes.data, colidx.ctypes.data, val.ctypes.data), "sblas_mm_read_csr")
    return r.value, c.value, z.value, s.value, rowptr, colidx[:z.value], val[:z.value]


def rand0to1(count, seed=211):
    """The reference's dense initialiser (DenseMatrix ctor, matrix.h:519-528): srand(seed), rand() / RAND_MAX."""
    out = np.empty(int(count), np.float64)
    check(lib().sblas_host_fill_rand0to1(out.ctypes.data, int(count), seed), "sblas_host_fill_rand0to1")
    return out


def find_row_of_nnz(rowptr, nnz_idx):
    rowptr = np.ascontiguousarray(rowptr, np.int32)
    return int(lib().sblas_find_row_of_nnz(rowptr.ctypes.data, len(rowptr) - 1, int(nnz_idx)))


def partition_nnz(rowptr, n_gpu, i_gpu):
    """-> dict(start_row, stop_row, nnz, first_nnz, rowptr) of GPU i's nnz-balanced row block."""
    rowptr = np.ascontiguousarray(rowptr, np.int32)
    rows = len(rowptr) - 1
    nnz = int(rowptr[-1])
    s, e, k, f = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    buf = np.zeros(rows + 2, np.int32)
    num = lib().sblas_partition_nnz(rowptr.ctypes.data, rows, nnz, n_gpu, i_gpu, C.byref(s), C.byref(e),
                                    C.byref(k), C.byref(f), buf.ctypes.data)
    if num < 0:
        raise SblasError("sblas_partition_nnz failed (%d)" % num)
    return dict(start_row=s.value, stop_row=e.value, nnz=k.value, first_nnz=f.value, rowptr=buf[:num].copy())


def partition_dense(first_order, n_gpu, i_gpu):
    o, d = C.c_int64(), C.c_int64()
    check(lib().sblas_partition_dense(first_order, n_gpu, i_gpu, C.byref(o), C.byref(d)), "sblas_partition_dense")
    return o.value, d.value


# ------------------------------------------------------------------------------------------
# device calls (torch tensors carry the device pointers)
# ------------------------------------------------------------------------------------------
def _dev_ptr(t, dtype, what):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise SblasError("%s must be a GPU tensor (no CPU path exists)" % what)
    if t.dtype != dtype or not t.is_contiguous():
        raise SblasError("%s must be a contiguous %s tensor" % (what, dtype))
    return t.data_ptr()


def _stream(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def spmm_workspace_bytes(rows, cols, nnz, n):
    return int(lib().sblas_hip_spmm_csr_f64_i32_workspace(rows, cols, nnz, n))


def spmm(rows, cols, rowptr, colidx, val, B, ldb, n, alpha, beta, Cmat, ldc, workspace, stream=None, c_offset=0):
    """C = alpha*A*B + beta*C through sblas_hip_spmm_csr_f64_i32.  B, C column-major flat tensors.
    c_offset: element offset into Cmat (method 2 points C at Ccopy + start_row)."""
    import torch
    nnz = int(colidx.numel())
    rc = lib().sblas_hip_spmm_csr_f64_i32(
        -1, _stream(stream), rows, cols, nnz,
        _dev_ptr(rowptr, torch.int32, "rowptr"), _dev_ptr(colidx, torch.int32, "colidx") if nnz else None,
        _dev_ptr(val, torch.float64, "val") if nnz else None,
        _dev_ptr(B, torch.float64, "B") if cols else None, ldb, n, alpha, beta,
        _dev_ptr(Cmat, torch.float64, "C") + 8 * c_offset, ldc,
        _dev_ptr(workspace, torch.float64, "workspace") if workspace is not None and workspace.numel() else None,
        workspace.numel() * 8 if workspace is not None else 0)
    check(rc, "sblas_hip_spmm_csr_f64_i32")


class SpmmPlan:
    """A per-matrix plan (sblas_hip_spmm_plan_create): the panel verdicts of one structure (rowptr, colidx) at one
    width n, taken once.  Keeps the structure tensors alive; destroy() / garbage collection frees the device buffer."""

    def __init__(self, rows, cols, rowptr, colidx, n, stream=None):
        import torch
        self.rows, self.cols, self.n, self.rowptr, self.colidx = rows, cols, n, rowptr, colidx
        self.nnz = int(colidx.numel())
        h = C.c_void_p()
        check(lib().sblas_hip_spmm_plan_create(-1, _stream(stream), rows, cols, self.nnz, _dev_ptr(rowptr, torch.int32, "rowptr"),
                                               _dev_ptr(colidx, torch.int32, "colidx") if self.nnz else None, n, C.byref(h)),
              "sblas_hip_spmm_plan_create")
        self.handle = h

    def info(self):
        out = (C.c_int64 * 8)()
        check(lib().sblas_hip_spmm_plan_info(self.handle, out), "sblas_hip_spmm_plan_info")
        return dict(active=bool(out[0]), windowed=int(out[1]), direct=int(out[2]), mfma=int(out[3]), merge=out[4] == 1,
                    four_rows=out[4] == 2,
                    stage_range=bool(out[5]), ldbt=int(out[6]), panel_rows=int(out[7]))

    def spmm(self, val, B, ldb, n, alpha, beta, Cmat, ldc, workspace, stream=None, c_offset=0):
        """The planned form of spmm(): same arguments, same results, only the kernels that have panels are launched."""
        import torch
        rc = lib().sblas_hip_spmm_csr_f64_i32_planned(
            self.handle, -1, _stream(stream), self.rows, self.cols, self.nnz,
            _dev_ptr(self.rowptr, torch.int32, "rowptr"), _dev_ptr(self.colidx, torch.int32, "colidx") if self.nnz else None,
            _dev_ptr(val, torch.float64, "val") if self.nnz else None,
            _dev_ptr(B, torch.float64, "B") if self.cols else None, ldb, n, alpha, beta,
            _dev_ptr(Cmat, torch.float64, "C") + 8 * c_offset, ldc,
            _dev_ptr(workspace, torch.float64, "workspace") if workspace is not None and workspace.numel() else None,
            workspace.numel() * 8 if workspace is not None else 0)
        check(rc, "sblas_hip_spmm_csr_f64_i32_planned")

    def destroy(self):
        if self.handle:
            lib().sblas_hip_spmm_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def dense_to_rowmajor(cols, n, B, ldb, Bt, stream=None):
    import torch
    ldbt = int(lib().sblas_hip_spmm_ldbt(n))
    check(lib().sblas_hip_dense_to_rowmajor_f64(-1, _stream(stream), cols, n, _dev_ptr(B, torch.float64, "B"), ldb,
                                                _dev_ptr(Bt, torch.float64, "Bt"), ldbt), "sblas_hip_dense_to_rowmajor_f64")
    return ldbt


def spmm_rowmajorB(rows, cols, rowptr, colidx, val, Bt, n, alpha, beta, Cmat, ldc, stream=None, c_offset=0):
    import torch
    nnz = int(colidx.numel())
    ldbt = int(lib().sblas_hip_spmm_ldbt(n))
    rc = lib().sblas_hip_spmm_csr_rowmajorB_f64_i32(
        -1, _stream(stream), rows, cols, nnz,
        _dev_ptr(rowptr, torch.int32, "rowptr"), _dev_ptr(colidx, torch.int32, "colidx") if nnz else None,
        _dev_ptr(val, torch.float64, "val") if nnz else None,
        _dev_ptr(Bt, torch.float64, "Bt") if Bt is not None else None, ldbt, n, alpha, beta,
        _dev_ptr(Cmat, torch.float64, "C") + 8 * c_offset, ldc)
    check(rc, "sblas_hip_spmm_csr_rowmajorB_f64_i32")


def validate_csr(rows, cols, rowptr, colidx, stream=None):
    """True when the CSR structure on the device is well formed (sblas_hip_debug_validate_csr_i32; synchronises)."""
    import torch
    nnz = int(colidx.numel())
    rc = lib().sblas_hip_debug_validate_csr_i32(-1, _stream(stream), rows, cols, nnz, _dev_ptr(rowptr, torch.int32, "rowptr"),
                                                _dev_ptr(colidx, torch.int32, "colidx") if nnz else None)
    if rc not in (0, 1):
        check(rc, "sblas_hip_debug_validate_csr_i32")
    return rc == 0


def panel_stats(reset=True):
    """(windowed, direct, fallback) panel counts of the SpMM launches since the last reset."""
    out = (C.c_uint64 * 4)()
    check(lib().sblas_hip_debug_spmm_panel_stats(out, 1 if reset else 0), "sblas_hip_debug_spmm_panel_stats")
    return int(out[0]), int(out[1]), int(out[2])


def kernel_events(enable):
    check(lib().sblas_hip_debug_spmm_kernel_events(1 if enable else 0), "sblas_hip_debug_spmm_kernel_events")


def last_kernel_ms():
    """Duration of the dominant stage-2 kernel of the most recent SpMM launch (waits for it); needs kernel_events(True)."""
    ms = C.c_float()
    check(lib().sblas_hip_debug_spmm_last_kernel_ms(C.byref(ms)), "sblas_hip_debug_spmm_last_kernel_ms")
    return float(ms.value)


def reload_env():
    """Have the library re-read its SBLAS_* experiment switches (it reads the environment only once)."""
    check(lib().sblas_hip_debug_reload_env(), "sblas_hip_debug_reload_env")


def spmv(rows, cols, rowptr, colidx, val, x, alpha, beta, y, stream=None, y_offset=0):
    import torch
    nnz = int(colidx.numel())
    rc = lib().sblas_hip_spmv_csr_f64_i32(
        -1, _stream(stream), rows, cols, nnz,
        _dev_ptr(rowptr, torch.int32, "rowptr"), _dev_ptr(colidx, torch.int32, "colidx") if nnz else None,
        _dev_ptr(val, torch.float64, "val") if nnz else None,
        _dev_ptr(x, torch.float64, "x"), alpha, beta, _dev_ptr(y, torch.float64, "y") + 8 * y_offset)
    check(rc, "sblas_hip_spmv_csr_f64_i32")


def merge_rowblocks_local(M, N, starts, nrows, blocks, alpha, beta, Cmat, ldc=None, stream=None):
    """C = beta*C + alpha * (packed row blocks scattered into place), sblas_hip_merge_rowblocks_local_f64.
    blocks[q]: flat float64 device tensor holding an nrows[q] x N column-major block (leading dimension nrows[q])."""
    import torch
    g = len(blocks)
    st = (C.c_int64 * g)(*[int(v) for v in starts])
    nr = (C.c_int64 * g)(*[int(v) for v in nrows])
    for q in range(g):
        if blocks[q] is not None and blocks[q].numel() < int(nrows[q]) * N:
            raise SblasError("block %d is smaller than nrows x N" % q)
    ptrs = (C.c_void_p * g)(*[_dev_ptr(b, torch.float64, "block") if b is not None and b.numel() else None for b in blocks])
    pc = _dev_ptr(Cmat, torch.float64, "C")
    check(lib().sblas_hip_merge_rowblocks_local_f64(-1, _stream(stream), M, N, g, st, nr, ptrs, alpha, beta, pc,
                                                    M if ldc is None else ldc), "sblas_hip_merge_rowblocks_local_f64")


def axpby(n, alpha, x, beta, y, stream=None):
    """y = beta*y + alpha*x (kernel.h:27-38 semantics)."""
    import torch
    px, py = _dev_ptr(x, torch.float64, "x"), _dev_ptr(y, torch.float64, "y")
    check(lib().sblas_hip_axpby_f64(-1, _stream(stream), n, alpha, px, beta, py), "sblas_hip_axpby_f64")


# ------------------------------------------------------------------------------------------
# one process driving several GPUs (the reference's process model): comm.hip through the C ABI
# ------------------------------------------------------------------------------------------
def comm_get(devs):
    """Persistent communicator set over the device list `devs` (sblas_hip_comm_get).  All equal = ranks folded onto one
    device (on-device sum / no copies); all distinct = RCCL over xGMI."""
    arr = (C.c_int * len(devs))(*[int(d) for d in devs])
    out = C.c_void_p()
    check(lib().sblas_hip_comm_get(len(devs), arr, C.byref(out)), "sblas_hip_comm_get")
    return out


def _ptr_array(tensors, what, allow_none=False):
    import torch
    vals = []
    for t in tensors:
        if t is None or (allow_none and t.numel() == 0):
            vals.append(None)
        else:
            vals.append(_dev_ptr(t, torch.float64, what))
    return (C.c_void_p * len(vals))(*vals)


def _stream_array(streams):
    return (C.c_void_p * len(streams))(*[s.cuda_stream for s in streams])


def allreduce_sum(comm, bufs, streams, count):
    """In-place sum of bufs[r] (rank r's buffer on rank r's device / stream): sblas_hip_allreduce_sum_f64."""
    check(lib().sblas_hip_allreduce_sum_f64(comm, _ptr_array(bufs, "buf"), _stream_array(streams), count),
          "sblas_hip_allreduce_sum_f64")


def merge_rowblocks(comm, M, N, starts, nrows, partial, gather, alpha, beta, Cs, ldc, streams):
    """Method-2 / SpMV merge over the ranks of `comm`: exchange the packed row blocks (RCCL send/recv; folded ranks
    skip the copies) and scatter + alpha/beta on every rank (sblas_hip_merge_rowblocks_f64)."""
    g = len(partial)
    st = (C.c_int64 * g)(*[int(v) for v in starts])
    nr = (C.c_int64 * g)(*[int(v) for v in nrows])
    ga = _ptr_array(gather, "gather", allow_none=True) if gather is not None else None
    check(lib().sblas_hip_merge_rowblocks_f64(comm, M, N, st, nr, _ptr_array(partial, "partial", allow_none=True), ga,
                                              alpha, beta, _ptr_array(Cs, "C"), ldc, _stream_array(streams)),
          "sblas_hip_merge_rowblocks_f64")


def panel_census(reset=True):
    """dict(windowed, direct, fallback, mfma): row panels per stage-2 kernel since the last reset."""
    out = (C.c_uint64 * 4)()
    check(lib().sblas_hip_debug_spmm_panel_stats(out, 1 if reset else 0), "sblas_hip_debug_spmm_panel_stats")
    return dict(windowed=int(out[0]), direct=int(out[1]), fallback=int(out[2]), mfma=int(out[3]))


# ------------------------------------------------------------------------------------------
# the other value / index types of the reference's templates (typed entry points; tags from the tensors' dtypes)
# ------------------------------------------------------------------------------------------
F64, F32, I32, I64 = 0, 1, 0, 1


def _tags(val_dtype, idx_dtype):
    import torch
    vt = {torch.float64: F64, torch.float32: F32}.get(val_dtype)
    it = {torch.int32: I32, torch.int64: I64}.get(idx_dtype)
    if vt is None or it is None:
        raise SblasError("values must be float32 / float64 and indices int32 / int64 tensors")
    return vt, it


def spmm_typed_workspace_bytes(val_dtype, idx_dtype, rows, cols, nnz, n):
    vt, it = _tags(val_dtype, idx_dtype)
    return int(lib().sblas_hip_spmm_csr_workspace(vt, it, rows, cols, nnz, n))


def spmm_typed(rows, cols, rowptr, colidx, val, B, ldb, n, alpha, beta, Cmat, ldc, workspace, stream=None, c_offset=0):
    """sblas_hip_spmm_csr: C = alpha*A*B + beta*C in the tensors' own value / index types (workspace: a uint8 tensor)."""
    import torch
    vt, it = _tags(val.dtype, rowptr.dtype)
    nnz = int(colidx.numel())
    rc = lib().sblas_hip_spmm_csr(
        -1, _stream(stream), vt, it, rows, cols, nnz, _dev_ptr(rowptr, rowptr.dtype, "rowptr"),
        _dev_ptr(colidx, rowptr.dtype, "colidx") if nnz else None, _dev_ptr(val, val.dtype, "val") if nnz else None,
        _dev_ptr(B, val.dtype, "B") if cols else None, ldb, n, alpha, beta,
        _dev_ptr(Cmat, val.dtype, "C") + Cmat.element_size() * c_offset, ldc,
        _dev_ptr(workspace, torch.uint8, "workspace") if workspace is not None and workspace.numel() else None,
        workspace.numel() if workspace is not None else 0)
    check(rc, "sblas_hip_spmm_csr")


def spmv_typed(rows, cols, rowptr, colidx, val, x, alpha, beta, y, stream=None, y_offset=0):
    vt, it = _tags(val.dtype, rowptr.dtype)
    nnz = int(colidx.numel())
    rc = lib().sblas_hip_spmv_csr(
        -1, _stream(stream), vt, it, rows, cols, nnz, _dev_ptr(rowptr, rowptr.dtype, "rowptr"),
        _dev_ptr(colidx, rowptr.dtype, "colidx") if nnz else None, _dev_ptr(val, val.dtype, "val") if nnz else None,
        _dev_ptr(x, val.dtype, "x"), alpha, beta, _dev_ptr(y, val.dtype, "y") + y.element_size() * y_offset)
    check(rc, "sblas_hip_spmv_csr")


def axpby_typed(n, alpha, x, beta, y, stream=None):
    vt, _ = _tags(x.dtype, __import__("torch").int32)
    check(lib().sblas_hip_axpby(-1, _stream(stream), vt, n, alpha, _dev_ptr(x, x.dtype, "x"), beta, _dev_ptr(y, x.dtype, "y")),
          "sblas_hip_axpby")


def _typed_ptr_array(tensors, dtype, what, allow_none=False):
    vals = []
    for t in tensors:
        vals.append(None if t is None or (allow_none and t.numel() == 0) else _dev_ptr(t, dtype, what))
    return (C.c_void_p * len(vals))(*vals)


def allreduce_sum_typed(comm, bufs, streams, count):
    vt, _ = _tags(bufs[0].dtype, __import__("torch").int32)
    check(lib().sblas_hip_allreduce_sum(comm, vt, _typed_ptr_array(bufs, bufs[0].dtype, "buf"), _stream_array(streams), count),
          "sblas_hip_allreduce_sum")


def merge_rowblocks_typed(comm, M, N, starts, nrows, partial, gather, alpha, beta, Cs, ldc, streams):
    dt = Cs[0].dtype
    vt, _ = _tags(dt, __import__("torch").int32)
    g = len(partial)
    st = (C.c_int64 * g)(*[int(v) for v in starts])
    nr = (C.c_int64 * g)(*[int(v) for v in nrows])
    ga = _typed_ptr_array(gather, dt, "gather", allow_none=True) if gather is not None else None
    check(lib().sblas_hip_merge_rowblocks(comm, vt, M, N, st, nr, _typed_ptr_array(partial, dt, "partial", allow_none=True), ga,
                                          alpha, beta, _typed_ptr_array(Cs, dt, "C"), ldc, _stream_array(streams)),
          "sblas_hip_merge_rowblocks")


def partition_nnz_i64(rowptr, n_gpu, i_gpu):
    """sblas_partition_nnz for 64-bit row pointers."""
    rowptr = np.ascontiguousarray(rowptr, np.int64)
    rows = len(rowptr) - 1
    s, e, k, f = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    buf = np.zeros(rows + 2, np.int64)
    num = lib().sblas_partition_nnz_i64(rowptr.ctypes.data, rows, int(rowptr[-1]), n_gpu, i_gpu, C.byref(s), C.byref(e),
                                        C.byref(k), C.byref(f), buf.ctypes.data)
    if num < 0:
        raise SblasError("sblas_partition_nnz_i64 failed (%d)" % num)
    return dict(start_row=s.value, stop_row=e.value, nnz=k.value, first_nnz=f.value, rowptr=buf[:num].copy())
