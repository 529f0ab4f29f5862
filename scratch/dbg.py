import sys, os, numpy as np, torch
sys.path.insert(0, 's-blas_amd/python'); sys.path.insert(0,'oracle')
import sblas_amd as S
from sblas_amd import synth
dev = torch.device('cuda:0')
for (rows, per, half) in ((1000, 40, 100), (4000, 399, 2000)):
    rp, ci, v = synth.banded(rows, per, half)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    n = 64
    B = torch.rand(rows*n, dtype=torch.float64, device=dev); C = torch.ones(rows*n, dtype=torch.float64, device=dev)
    ws = torch.empty(rows*64, dtype=torch.float64, device=dev)
    S.panel_stats()
    S.spmm(rows, rows, d(rp), d(ci), d(v), B, rows, n, 1.0, 1.0, C, rows, ws)
    torch.cuda.synchronize()
    print(rows, per, half, 'stats (windowed, direct, fallback):', S.panel_stats())
