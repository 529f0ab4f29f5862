"""BASELINE config 5 shape at reduced size on one GPU: Queen_4147-like rows (3-D FEM stencil clusters, ~66 nnz/row,
band +-50000), N = 256 -> every panel goes to the direct DPP kernel (128-column tiles, two column chunks).
Prints time, GFLOP/s and the algorithmic HBM rate; checks 32 rows against the oracle."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sblas_amd as S
from sblas_amd import synth
import oracle_py as O
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
t0 = time.time(); rp, ci, v = synth.queen_like(rows, progress=(lambda r: print("  generated %d rows" % r, flush=True) if r % (16 * 65536) == 0 else None)); print("generated %d rows, %d nnz in %.1f s" % (rows, len(ci), time.time() - t0), flush=True)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rowptr, colidx, val = d(rp), d(ci), d(v)
Bh = torch.rand(rows * n, dtype=torch.float64, generator=torch.Generator().manual_seed(211))
B, C = Bh.to(dev), torch.ones(rows * n, dtype=torch.float64, device=dev)
ws = torch.empty(S.spmm_workspace_bytes(rows, rows, len(ci), n) // 8, dtype=torch.float64, device=dev)
S.panel_stats()
for _ in range(2): S.spmm(rows, rows, rowptr, colidx, val, B, rows, n, 1.0, 1.0, C, rows, ws)
st = S.panel_stats()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
steps = 5
e0.record()
for _ in range(steps): S.spmm(rows, rows, rowptr, colidx, val, B, rows, n, 1.0, 1.0, C, rows, ws)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
nnz = len(ci)
alg = nnz * 12 + (rows + 1) * 4 + 8 * rows * n + 16 * rows * n
ref = np.zeros(rows * n); r0 = rows // 2
O.spmm_rows(r0, r0 + 32, rows, rows, n, rp, ci, v, Bh.numpy(), ref, 1.0, 0.0)
got = C.view(n, rows)[:, r0:r0 + 32].cpu().numpy(); want = 1.0 + (2 + steps) * ref.reshape(n, rows)[:, r0:r0 + 32]
print("panels (windowed, direct, fallback):", st)
print("queen-like %d rows N=%d: %.3f ms/step  %.1f GFLOP/s  alg %.1f GB/s (%.1f %% of 8 TB/s)  oracle %s" %
      (rows, n, ms, 2.0 * nnz * n / ms / 1e6, alg / ms / 1e6, alg / ms / 1e6 / 80.0, np.allclose(got, want, rtol=1e-9, atol=1e-9)))
