"""Randomised differential test of the C-ABI SpMM / SpMV entry points against the oracle (one GPU).

  python tools/fuzz_parity.py [--cases N] [--seed S] [--max-rows R]

Every case draws a matrix family (banded / scattered / power-law / grid-structured / block-structured / random with
empty and very long rows), optional damage (unsorted rows, duplicate columns), a row block of it (method-2 style:
re-based row pointers, C at an offset, ldc > rows), a column count, leading dimensions, alpha / beta (including 0 and
NaN-filled C with beta = 0), a kernel-selection switch, -- one fp64 / int32 case in two -- the planned form of the call (bit-identical
to the unplanned one) and -- one case in four -- one of the other value / index types.
fp64: 1e-10 relative (north_star), fp32: 1e-4.  Exit status 1 on the first mismatch (the case's parameters are printed,
`--seed S --only K` replays it)."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sblas_amd as S
from sblas_amd import synth
import oracle_py as O

VARIANTS = ["auto", "auto", "auto", "dpp", "rows", "merge", "mfma", "nomfma"]
SWITCHES = ["SBLAS_SPMM_VARIANT", "SBLAS_STAGE_RANGE", "SBLAS_SPMV_VARIANT"]


def family(rng, max_rows):
    kind = rng.choice(["banded", "scattered", "powerlaw", "grid", "blocks", "random"])
    rows = int(rng.integers(1, max_rows))
    if kind == "banded":
        per = int(rng.integers(1, 200))
        rp, ci, v = synth.banded(rows, per, int(rng.integers(per, 40 * per + 2)), seed=int(rng.integers(1 << 30)))
        cols = rows
    elif kind == "scattered":
        rows = max(rows, 64)
        rp, ci, v = synth.queen_like(rows, seed=int(rng.integers(1 << 30)), half_band=int(rng.integers(50, max(51, rows))))
        cols = rows
    elif kind == "powerlaw":
        rp, ci, v = synth.powerlaw(max(rows, 10), avg=float(rng.uniform(1.5, 12)), max_len=int(rng.integers(10, 3000)), seed=int(rng.integers(1 << 30)))
        cols = rows = len(rp) - 1
    elif kind == "grid":
        rp, ci, v = synth.queen_like_grid(max(rows, 300), seed=int(rng.integers(1 << 30)), half_band=int(rng.integers(100, 3000)))
        cols = rows = len(rp) - 1
    elif kind == "blocks":
        rows = max(rows, 64)
        rp, ci, v = synth.block_structured(rows, nnz_per_row=int(rng.integers(8, 300)), half_band=int(rng.integers(300, 3000)),
                                           fill=float(rng.uniform(0.2, 1.0)), seed=int(rng.integers(1 << 30)))
        cols = rows
    else:
        cols = int(rng.integers(1, max_rows))
        long_row = (int(rng.integers(rows)), int(rng.integers(1, 4 * cols + 2))) if rng.random() < 0.5 else None
        rp, ci, v = synth.random_csr(rows, cols, float(rng.uniform(0.2, 60)), seed=int(rng.integers(1 << 30)),
                                     sorted_rows=bool(rng.random() < 0.5), empty_every=int(rng.choice([0, 0, 3, 17])), long_row=long_row)
    rp, ci, v = np.asarray(rp, np.int64), np.asarray(ci, np.int64), np.asarray(v, np.float64)
    damage = rng.choice(["none", "none", "none", "shuffle", "dups"])
    if damage == "shuffle" and len(ci):
        for r in rng.integers(0, rows, size=max(1, rows // 7)):
            a, b = rp[r], rp[r + 1]
            perm = rng.permutation(b - a)
            ci[a:b], v[a:b] = ci[a:b][perm], v[a:b][perm]
    elif damage == "dups" and len(ci) > 1:
        idx = rng.integers(1, len(ci), size=max(1, len(ci) // 50))
        ci[idx] = ci[idx - 1]                         # may cross a row boundary: still a valid column
    return kind + "/" + damage, rows, cols, rp, ci, v


def one_case(case, rng, dev, max_rows):
    while True:
        try:
            desc, rows, cols, rp, ci, v = family(rng, max_rows)
            break
        except (ValueError, AssertionError, IndexError, ZeroDivisionError):     # a parameter draw the generator refuses
            continue
    # row block
    if rng.random() < 0.5 and rows > 2:
        a = int(rng.integers(0, rows - 1))
        b = int(rng.integers(a + 1, rows + 1))
    else:
        a, b = 0, rows
    m = b - a
    sub_rp = rp[a:b + 1] - rp[a]
    sub_ci, sub_v = ci[rp[a]:rp[b]], v[rp[a]:rp[b]]
    n = int(rng.choice([1, 2, 5, 8, 9, 12, 16, 17, 24, 32, 33, 64, 65, 100, 128, 130, 200, 256, 300]))
    ldb = cols + int(rng.choice([0, 0, 3]))
    M_full = rows
    ldc = M_full + int(rng.choice([0, 0, 5]))
    alpha = float(rng.choice([1.0, -2.5, 0.0, 3.0]))
    beta = float(rng.choice([0.0, 1.0, 4.0, -0.5]))
    typed = rng.random() < 0.25
    vt = np.float32 if typed and rng.random() < 0.6 else np.float64
    it = np.int64 if typed and (vt == np.float64 or rng.random() < 0.5) else np.int32
    variant = str(rng.choice(VARIANTS))
    stage_range = str(rng.choice(["", "", "0", "1"]))
    for k in SWITCHES:
        os.environ.pop(k, None)
    if variant != "auto":
        os.environ["SBLAS_SPMM_VARIANT"] = variant
    if stage_range:
        os.environ["SBLAS_STAGE_RANGE"] = stage_range
    S.reload_env()
    params = dict(case=case, family=desc, rows=rows, cols=cols, block=(a, b), nnz=int(len(sub_ci)), n=n, ldb=ldb, ldc=ldc, alpha=alpha,
                  beta=beta, vt=np.dtype(vt).name, it=np.dtype(it).name, variant=variant, stage_range=stage_range)
    B = rng.standard_normal(ldb * n).astype(vt)
    C0 = rng.standard_normal(ldc * n).astype(vt)
    if beta == 0.0:
        C0[:] = np.nan                                   # beta = 0 must not read C ... inside the block's rows
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    drp, dci, dv = d(sub_rp.astype(it)), d(sub_ci.astype(it)), d(sub_v.astype(vt))
    dB, dC = d(B), d(C0.copy())
    if vt == np.float64 and it == np.int32 and rng.random() < 0.7:
        nbytes = S.spmm_workspace_bytes(m, cols, len(sub_ci), n)
        ws = torch.full((max(nbytes // 8, 1),), float("nan"), dtype=torch.float64, device=dev)
        S.spmm(m, cols, drp, dci, dv, dB, ldb, n, alpha, beta, dC, ldc, ws if nbytes else None, c_offset=a)
        if rng.random() < 0.5:
            # the planned form of the same call (sblas_hip_spmm_plan_*): bit-identical to the unplanned one
            plan = S.SpmmPlan(m, cols, drp, dci, n)
            dCp = d(C0.copy())
            plan.spmm(dv, dB, ldb, n, alpha, beta, dCp, ldc, ws if nbytes else None, c_offset=a)
            torch.cuda.synchronize()
            same = torch.equal(dC.view(torch.int64), dCp.view(torch.int64))
            params["plan"] = plan.info()
            plan.destroy()
            if not same:
                print("MISMATCH planned vs unplanned", params, flush=True)
                return False
    else:
        nbytes = S.spmm_typed_workspace_bytes(dv.dtype, drp.dtype, m, cols, len(sub_ci), n)
        ws = torch.full((max(nbytes, 1),), 0xFF, dtype=torch.uint8, device=dev)
        S.spmm_typed(m, cols, drp, dci, dv, dB, ldb, n, alpha, beta, dC, ldc, ws if nbytes else None, c_offset=a)
    torch.cuda.synchronize()
    got = dC.cpu().numpy().reshape(n, ldc)
    # oracle on packed arrays
    Bp = np.ascontiguousarray(B.reshape(n, ldb)[:, :cols]).reshape(-1)
    Cp = np.ascontiguousarray(C0.reshape(n, ldc)[:, a:b]).reshape(-1)
    if beta == 0.0:
        Cp = np.zeros_like(Cp)
    ref = O.spmm_typed(m, cols, n, sub_rp.astype(it), sub_ci.astype(it), sub_v.astype(vt), Bp, Cp, vt(alpha), vt(beta)).reshape(n, m)
    longest = int(np.diff(sub_rp).max()) if m else 0
    # fp32: the sum of a row of L terms carries ~ sqrt(L) * 6e-8 * |partial sums| in either summation order
    tol = dict(rtol=1e-4, atol=1e-4 * max(1.0, longest / 64.0)) if vt == np.float32 else dict(rtol=1e-10, atol=1e-11)
    ok = np.allclose(got[:, a:b], ref, **tol)
    C0m = C0.reshape(n, ldc)
    outside = np.ones(ldc, bool)
    outside[a:b] = False
    ok_out = np.array_equal(got[:, outside], C0m[:, outside], equal_nan=True)       # rows outside the block untouched
    # SpMV on the same block
    x = rng.standard_normal(cols).astype(vt)
    y0 = rng.standard_normal(M_full).astype(vt)
    dx, dy = d(x), d(y0.copy())
    if vt == np.float64 and it == np.int32:
        S.spmv(m, cols, drp, dci, dv, dx, alpha, beta if beta != 0.0 else 1.0, dy, y_offset=a)
    else:
        S.spmv_typed(m, cols, drp, dci, dv, dx, alpha, beta if beta != 0.0 else 1.0, dy, y_offset=a)
    torch.cuda.synchronize()
    bb = beta if beta != 0.0 else 1.0
    yref = O.spmv_typed(m, sub_rp.astype(it), sub_ci.astype(it), sub_v.astype(vt), x, y0[a:b].copy(), vt(alpha), vt(bb))
    goty = dy.cpu().numpy()
    ok_v = np.allclose(goty[a:b], yref, **tol) and np.array_equal(goty[outside[:M_full]], y0[outside[:M_full]])
    if not (ok and ok_out and ok_v):
        worst = float(np.nanmax(np.abs(got[:, a:b] - ref))) if m and n else 0.0
        print("MISMATCH", params, "spmm", ok, "untouched", ok_out, "spmv", ok_v, "max abs diff", worst, flush=True)
        return False
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-rows", type=int, default=6000)
    ap.add_argument("--only", type=int, default=-1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    bad = 0
    for case in range(args.cases):
        rng = np.random.default_rng([args.seed, case])
        if args.only >= 0 and case != args.only:
            continue
        if not one_case(case, rng, dev, args.max_rows):
            bad += 1
            break
        if case % 25 == 24:
            print("%d cases ok" % (case + 1), flush=True)
    for k in SWITCHES:
        os.environ.pop(k, None)
    print("fuzz: %s" % ("FAILED" if bad else "all cases match the oracle"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
