"""SpMV on Queen_4147-like rows (69 nnz/row, band +-50000) on one GPU: time, GB/s, oracle check."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sblas_amd as S
from sblas_amd import synth
import oracle_py as O
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
dev = torch.device("cuda:0")
rp, ci, v = synth.queen_like(rows)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rowptr, colidx, val = d(rp), d(ci), d(v)
xh = np.random.default_rng(1).standard_normal(rows)
x, y = d(xh), torch.zeros(rows, dtype=torch.float64, device=dev)
for _ in range(3): S.spmv(rows, rows, rowptr, colidx, val, x, 1.0, 0.0, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): S.spmv(rows, rows, rowptr, colidx, val, x, 1.0, 0.0, y)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
nnz = len(ci); alg = nnz * 12 + (rows + 1) * 4 + 8 * rows + 16 * rows
ok = np.allclose(y.cpu().numpy(), O.spmv(rows, rp, ci, v, xh, np.zeros(rows), 1.0, 0.0), rtol=1e-10, atol=1e-10)
print("queen-like SpMV %d rows, %d nnz (%.1f/row), variant %s: %.1f us  %.0f GB/s (%.1f %% of 8 TB/s)  oracle %s" %
      (rows, nnz, nnz / rows, os.environ.get("SBLAS_SPMV_VARIANT", "auto"), us, alg / us / 1e3, alg / us / 80e3, ok))
