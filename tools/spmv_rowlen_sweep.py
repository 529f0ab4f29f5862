"""SpMV variants by row length (banded synthetic, 600k rows, band +-20000 or argv[3]): us per call for each variant."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python"))
import sblas_amd as S
from sblas_amd import synth
dev = torch.device("cuda:0")
rows = 600000
for per in [int(a) for a in sys.argv[1].split(",")]:
    rp, ci, v = synth.banded(rows, per, int(sys.argv[3]) if len(sys.argv) > 3 else 20000)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rowptr, colidx, val = d(rp), d(ci), d(v)
    x, y = torch.ones(rows, dtype=torch.float64, device=dev), torch.zeros(rows, dtype=torch.float64, device=dev)
    alg = len(ci) * 12 + rows * 28
    out = []
    for var in sys.argv[2].split(","):
        os.environ["SBLAS_SPMV_VARIANT"] = var; S.reload_env()
        for _ in range(3): S.spmv(rows, rows, rowptr, colidx, val, x, 1.0, 0.0, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): S.spmv(rows, rows, rowptr, colidx, val, x, 1.0, 0.0, y)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        out.append("%s %.0f us (%.2f TB/s)" % (var, us, alg / us / 1e6))
    print("nnz/row %3d: " % per + " | ".join(out), flush=True)
