"""One SpMV shape for counter runs: python tools/spmv_one.py <per_row> <half_band> <rows> [launches] (banded synthetic)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python"))
import sblas_amd as S
from sblas_amd import synth
per, hb, rows = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
rp, ci, v = synth.banded(rows, per, hb)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rowptr, colidx, val = d(rp), d(ci), d(v)
x, y = torch.ones(rows, dtype=torch.float64, device=dev), torch.zeros(rows, dtype=torch.float64, device=dev)
for _ in range(k): S.spmv(rows, rows, rowptr, colidx, val, x, 1.0, 1.0, y)
torch.cuda.synchronize()
print("nnz %d algorithmic bytes %d" % (len(ci), len(ci) * 12 + (rows + 1) * 4 + 8 * rows + 16 * rows))
