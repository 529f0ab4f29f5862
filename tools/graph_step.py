"""Bench-matrix SpMM step, eager launches against a captured HIP graph of the same step (launch gaps)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python"))
import sblas_amd as S
from sblas_amd import synth
dev = torch.device("cuda:0")
rows, n = 72000, 64
rows, (rp, ci, v) = synth.nd24k_like()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rowptr, colidx, val = d(rp), d(ci), d(v)
B = torch.rand(rows * n, dtype=torch.float64, device=dev); C = torch.ones(rows * n, dtype=torch.float64, device=dev)
ws = torch.empty(S.spmm_workspace_bytes(rows, rows, len(ci), n) // 8, dtype=torch.float64, device=dev)
step = lambda: S.spmm(rows, rows, rowptr, colidx, val, B, rows, n, 1.0, 1.0, C, rows, ws)
for _ in range(5): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
def timed(f, k=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
for rep in range(3):
    print("eager %.4f ms | graph %.4f ms" % (timed(step), timed(g.replay)), flush=True)
# clock ramp: ms per step over consecutive blocks of 10 steps after 2 s of idling
import time
time.sleep(2.0)
print("after idle: " + " ".join("%.3f" % timed(step, 10) for _ in range(30)), flush=True)
time.sleep(2.0)
print("after idle: " + " ".join("%.3f" % timed(step, 10) for _ in range(30)), flush=True)
