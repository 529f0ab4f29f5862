"""SpMM / SpMV on short-row matrices (banded, 5..40 nonzeros per row): where does the direct path stand?"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python"))
import sblas_amd as S
from sblas_amd import synth
dev = torch.device("cuda:0")
rows, n = 1000000, 64
for per in (5, 10, 20, 40):
    rp, ci, v = synth.banded(rows, per, 5000)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rowptr, colidx, val = d(rp), d(ci), d(v)
    B = torch.rand(rows * n, dtype=torch.float64, device=dev); C = torch.zeros(rows * n, dtype=torch.float64, device=dev)
    ws = torch.empty(S.spmm_workspace_bytes(rows, rows, len(ci), n) // 8, dtype=torch.float64, device=dev)
    for _ in range(2): S.spmm(rows, rows, rowptr, colidx, val, B, rows, n, 1.0, 0.0, C, rows, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): S.spmm(rows, rows, rowptr, colidx, val, B, rows, n, 1.0, 0.0, C, rows, ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nnz = len(ci); alg = nnz * 12 + rows * 4 + 8 * rows * n + 8 * rows * n
    print("banded %d/row, 1M rows, N=64: %.3f ms  %.0f GFLOP/s  alg %.0f GB/s  panels %s" % (per, ms, 2.0 * nnz * n / ms / 1e6, alg / ms / 1e6, S.panel_stats()), flush=True)
