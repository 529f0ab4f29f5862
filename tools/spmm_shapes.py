"""One-GPU timing of the SpMM path on the stand-in shapes, several kernel selections side by side (interleaved rounds in
one process, median of the rounds), with the panel census and an oracle check of 32 rows per shape.

  python tools/spmm_shapes.py SHAPE[:ARG[:ARG]] ... [--n N] [--variants auto,nomfma,dpp] [--rounds R] [--steps K]
    nd24k[:scale]              banded uniform stand-in (bench shape)
    blocks[:rows[:fill]]       nd24k-like rows in dense 16 x 4 sub-blocks (synth.block_structured)
    blocksw[:rows[:fill]]      16 x 4 sub-blocks, 80 per row, over a +-50 000 band (direct-class panels)
    queen[:rows]               Queen-like, 40 scattered offsets (synth.queen_like)
    qgrid[:rows[:dofs[:hb]]]   Queen-like on a structured 3-D grid (synth.queen_like_grid), 3 unknowns per node and a +-50 000 band unless given
    powerlaw[:rows[:avg[:max]]] webbase-like row lengths (synth.powerlaw)
    uniform:rows:avg           binomial row lengths, columns anywhere (synth.random_csr)
    stencil:rows:noff:hb:lo:hi:cl  every row picks lo..hi of the same noff offsets (+-hb), each a run of cl columns
Environment switches of the library can be set per variant as name=ENV1=val1+ENV2=val2."""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sblas_amd as S
from sblas_amd import synth
import oracle_py as O


def make(shape):
    parts = shape.split(":")
    kind = parts[0]
    if kind == "nd24k":
        rows, (rp, ci, v) = synth.nd24k_like(float(parts[1]) if len(parts) > 1 else 1.0)
    elif kind == "blocks":
        rows = int(parts[1]) if len(parts) > 1 else 72000
        rp, ci, v = synth.block_structured(rows, fill=float(parts[2]) if len(parts) > 2 else 0.6)
    elif kind == "blocksw":   # the same sub-blocks scattered over a +-50 000 band: no panel fits the LDS-tiled kernel
        rows = int(parts[1]) if len(parts) > 1 else 300000
        rp, ci, v = synth.block_structured(rows, nnz_per_row=80, half_band=50000, fill=float(parts[2]) if len(parts) > 2 else 0.6)
    elif kind == "queen":
        rows = int(parts[1]) if len(parts) > 1 else 300000
        rp, ci, v = synth.queen_like(rows)
    elif kind == "qgrid":
        rp, ci, v = synth.queen_like_grid(int(parts[1]) if len(parts) > 1 else 300000, dofs=int(parts[2]) if len(parts) > 2 else 3,
                                          half_band=int(parts[3]) if len(parts) > 3 else 50000)
        rows = len(rp) - 1
    elif kind == "banded":
        rows = int(parts[1])
        rp, ci, v = synth.banded(rows, int(parts[2]), int(parts[3]) if len(parts) > 3 else 2000)
    elif kind == "uniform":   # binomial row lengths around the average, columns anywhere (synth.random_csr)
        rows = int(parts[1])
        rp, ci, v = synth.random_csr(rows, rows, float(parts[2]), sorted_rows=len(parts) > 3)
    elif kind == "powerlaw":
        rows = int(parts[1]) if len(parts) > 1 else 1000000
        rp, ci, v = synth.powerlaw(rows, avg=float(parts[2]) if len(parts) > 2 else 3.0, max_len=int(parts[3]) if len(parts) > 3 else 5000)
    elif kind == "stencil":   # stencil:rows:noff:half_band:lo:hi:cluster -- every row picks lo..hi of the same noff offsets,
        rows, noff, hb, lo, hi, cl = (int(x) for x in parts[1:7])   # each a run of `cluster` consecutive columns
        rng = np.random.Generator(np.random.MT19937(7))
        offs = np.unique(np.concatenate([[0], (rng.integers(-hb, hb + 1, noff) // cl) * cl]))
        k = rng.integers(lo, hi + 1, rows)
        rank = np.argsort(np.argsort(rng.random((rows, len(offs))), axis=1), axis=1)
        base = (np.arange(rows)[:, None] // cl) * cl + offs[None, :]
        mask = (rank < k[:, None]) & (base >= 0) & (base + cl <= rows)
        mask[:, np.searchsorted(offs, 0)] |= ~mask.any(1)
        rp = np.zeros(rows + 1, np.int64)
        np.cumsum(mask.sum(1) * cl, out=rp[1:])
        ci = (base[mask][:, None] + np.arange(cl)[None, :]).reshape(-1).astype(np.int32)
        rp = rp.astype(np.int32)
        v = rng.random(len(ci)) * 2.0 - 1.0
    else:
        raise SystemExit("unknown shape " + shape)
    return rows, rp, ci, v


def set_variant(spec):
    """'name' or 'name=ENV=val+ENV=val' -> environment for the library, then reload"""
    for k in [k for k in os.environ if k.startswith("SBLAS_") and k not in KEEP]:
        del os.environ[k]
    name, _, envs = spec.partition("=")
    if name not in ("auto", ""):
        os.environ["SBLAS_SPMM_VARIANT"] = name
    for kv in envs.split("+") if envs else []:
        k, _, val = kv.partition("=")
        os.environ[k] = val
    S.reload_env()


KEEP = {k for k in os.environ if k.startswith("SBLAS_")}     # switches set by the caller stay for every variant


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes", nargs="+")
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--variants", default="auto")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--beta", type=float, default=1.0, help="beta of the timed steps")
    ap.add_argument("--block", default="", help="i/g: time row block i of g (split by nonzeros, as method 2 does)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n = args.n
    variants = args.variants.split(",")
    for shape in args.shapes:
        t0 = time.time()
        rows, rp, ci, v = make(shape)
        K = rows
        if args.block:
            i, g = (int(x) for x in args.block.split("/"))
            cut = np.searchsorted(rp, [len(ci) * i // g, len(ci) * (i + 1) // g])
            a, b = int(cut[0]), max(int(cut[1]), int(cut[0]) + 1)
            ci, v = ci[rp[a]:rp[b]], v[rp[a]:rp[b]]
            rp = (rp[a:b + 1] - rp[a]).astype(np.int32)
            rows = b - a
            print("row block %s: rows [%d, %d) of %d" % (args.block, a, b, K))
        nnz = len(ci)
        print("%s: %d rows, %d nnz (%.1f per row), generated in %.1f s" % (shape, rows, nnz, nnz / rows, time.time() - t0), flush=True)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        rowptr, colidx, val = d(rp), d(ci), d(v)
        Bh = O.rand0to1(K * n)
        B = d(Bh)
        ws = torch.empty(S.spmm_workspace_bytes(rows, K, nnz, n) // 8, dtype=torch.float64, device=dev)
        alg = nnz * 12 + (rows + 1) * 4 + 8 * K * n + 16 * rows * n
        r0 = rows // 2
        ref = np.zeros(rows * n)
        O.spmm_rows(r0, r0 + 32, rows, K, n, rp, ci, v, Bh, ref, 1.0, 0.0)
        want = ref.reshape(n, rows)[:, r0:r0 + 32]
        times = {vs: [] for vs in variants}
        census, ok = {}, {}
        for rnd in range(args.rounds):
            for vs in variants:
                set_variant(vs)
                C = torch.zeros(rows * n, dtype=torch.float64, device=dev)
                S.panel_census()
                S.spmm(rows, K, rowptr, colidx, val, B, K, n, 1.0, 0.0, C, rows, ws)      # warm-up + check
                census[vs] = S.panel_census()
                got = C.view(n, rows)[:, r0:r0 + 32].cpu().numpy()
                ok[vs] = bool(np.allclose(got, want, rtol=1e-10, atol=1e-12))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.steps):
                    S.spmm(rows, K, rowptr, colidx, val, B, K, n, 1.0, args.beta, C, rows, ws)
                e1.record()
                torch.cuda.synchronize()
                times[vs].append(e0.elapsed_time(e1) / args.steps)
        for vs in variants:
            ms = float(np.median(times[vs]))
            print("  %-40s N=%d  %.4f ms/step (min %.4f)  %.0f GFLOP/s  alg %.0f GB/s = %.3f of 8 TB/s  panels %s  oracle %s" %
                  (vs, n, ms, min(times[vs]), 2.0 * nnz * n / ms / 1e6, alg / ms / 1e6, alg / ms / 1e6 / 8000.0, census[vs], ok[vs]), flush=True)
        del rowptr, colidx, val, B, ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
