"""One-GPU SpMV timing on stand-in shapes, several kernel selections side by side (interleaved rounds, median).
  python tools/spmv_shapes.py SHAPE ... [--variants auto,plain,stream]   (shapes as in tools/spmm_shapes.py)"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import sblas_amd as S
import oracle_py as O
ap = argparse.ArgumentParser()
ap.add_argument("shapes", nargs="+")
ap.add_argument("--variants", default="auto")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()

from spmm_shapes import make                    # noqa: E402
dev = torch.device("cuda:0")
for shape in args.shapes:
    rows, rp, ci, v = make(shape)
    nnz = len(ci)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rowptr, colidx, val = d(rp), d(ci), d(v)
    xh = np.random.default_rng(1).standard_normal(rows)
    x = d(xh)
    ref = O.spmv(rows, rp, ci, v, xh, np.zeros(rows), 1.0, 0.0)
    alg = nnz * 12 + (rows + 1) * 4 + 8 * rows + 16 * rows
    print("%s: %d rows, %d nnz, max row %d" % (shape, rows, nnz, int(np.diff(rp).max())), flush=True)
    times = {vs: [] for vs in args.variants.split(",")}
    ok = {}
    for rnd in range(args.rounds):
        for vs in times:
            os.environ["SBLAS_SPMV_VARIANT"] = vs
            S.reload_env()
            y = torch.zeros(rows, dtype=torch.float64, device=dev)
            S.spmv(rows, rows, rowptr, colidx, val, x, 1.0, 0.0, y)
            ok[vs] = bool(np.allclose(y.cpu().numpy(), ref, rtol=1e-10, atol=1e-12))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                S.spmv(rows, rows, rowptr, colidx, val, x, 1.0, 1.0, y)
            e1.record()
            torch.cuda.synchronize()
            times[vs].append(e0.elapsed_time(e1) / args.steps)
    for vs in times:
        ms = float(np.median(times[vs]))
        print("  %-10s %.4f ms  %.0f GB/s alg = %.3f of 8 TB/s  oracle %s" % (vs, ms, alg / ms / 1e6, alg / ms / 8e9, ok[vs]), flush=True)
