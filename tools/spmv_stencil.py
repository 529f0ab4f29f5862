"""SpMV on short, gather-friendly rows (dense narrow band = stencil-like), 4M rows: us, TB/s."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python"))
import sblas_amd as S
from sblas_amd import synth
dev = torch.device("cuda:0")
rows = 4000000
for per, hb in ((5, 4), (7, 6), (13, 12), (27, 40)):
    rp, ci, v = synth.banded(rows, per, hb)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rowptr, colidx, val = d(rp), d(ci), d(v)
    x, y = torch.ones(rows, dtype=torch.float64, device=dev), torch.zeros(rows, dtype=torch.float64, device=dev)
    alg = len(ci) * 12 + rows * 28
    out = []
    for var in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["auto"]):
        os.environ["SBLAS_SPMV_VARIANT"] = var; S.reload_env()
        for _ in range(3): S.spmv(rows, rows, rowptr, colidx, val, x, 1.0, 0.0, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): S.spmv(rows, rows, rowptr, colidx, val, x, 1.0, 0.0, y)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        out.append("%s %.0f us (%.2f TB/s)" % (var, us, alg / us / 1e6))
    print("%d per row, band +-%d, 4M rows: " % (per, hb) + " | ".join(out), flush=True)
