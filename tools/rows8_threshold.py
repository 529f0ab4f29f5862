"""N = 8 SpMM by row length (banded-random rows, 600k rows): lane-group kernel against the wave-per-row kernel
(SBLAS_ROWS8_MIN_AVG moves the switch-over; default 96)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python"))
import sblas_amd as S
from sblas_amd import synth
rows, n = 600000, 8
dev = torch.device("cuda:0"); d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for per in [int(a) for a in sys.argv[1].split(",")]:
    rp, ci, v = synth.banded(rows, per, 20000)
    rowptr, colidx, val = d(rp), d(ci), d(v)
    B = torch.rand(rows * n, dtype=torch.float64, device=dev); C = torch.zeros(rows * n, dtype=torch.float64, device=dev)
    ws = torch.empty(S.spmm_workspace_bytes(rows, rows, len(ci), n) // 8, dtype=torch.float64, device=dev)
    out = []
    for thr in ("100000", "1"):
        os.environ["SBLAS_ROWS8_MIN_AVG"] = thr; S.reload_env()
        for _ in range(3): S.spmm(rows, rows, rowptr, colidx, val, B, rows, n, 1.0, 0.0, C, rows, ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): S.spmm(rows, rows, rowptr, colidx, val, B, rows, n, 1.0, 0.0, C, rows, ws)
        e1.record(); torch.cuda.synchronize()
        out.append("%s %.3f ms" % ("lane groups" if thr != "1" else "wave per row", e0.elapsed_time(e1) / 20))
    print("banded %d/row N=8: " % per + " | ".join(out), flush=True)
