"""One timing of the typed entry points (fp32 values and / or int64 indices: typed_kernels.hip) on the bench matrix,
N = 64, next to the tuned <int32, fp64> pair reached through the same entry points.  python tools/typed_time.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, "s-blas_amd/python")
import sblas_amd as S
from sblas_amd import synth
dev = torch.device("cuda:0")
rows, (rp, ci, v) = synth.nd24k_like(1.0)
n = 64
for vt, it in ((np.float32, np.int32), (np.float64, np.int64), (np.float32, np.int64), (np.float64, np.int32)):
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    drp, dci, dv = d(rp.astype(it)), d(ci.astype(it)), d(v.astype(vt))
    B = d(np.random.default_rng(0).random(rows * n).astype(vt)); C = torch.zeros(rows * n, dtype=dv.dtype, device=dev)
    x = d(np.random.default_rng(1).random(rows).astype(vt)); y = torch.zeros(rows, dtype=dv.dtype, device=dev)
    ws = torch.empty(max(S.spmm_typed_workspace_bytes(dv.dtype, drp.dtype, rows, rows, len(ci), n), 1), dtype=torch.uint8, device=dev)
    for op in ("spmm", "spmv"):
        f = (lambda: S.spmm_typed(rows, rows, drp, dci, dv, B, rows, n, 1.0, 1.0, C, rows, ws)) if op == "spmm" else (lambda: S.spmv_typed(rows, rows, drp, dci, dv, x, 1.0, 1.0, y))
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        fl = 2.0 * len(ci) * (n if op == "spmm" else 1)
        print("%s %s/%s: %.3f ms  %.0f GFLOP/s" % (op, np.dtype(vt).name, np.dtype(it).name, ms, fl / ms / 1e6), flush=True)
