"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.
Tolerance: fp64 within 1e-10 relative (north_star); indices/partition tables are integer-exact and are
checked in the CPU suite.  Everything here is @pytest.mark.gpu and fails (not skips) without a device."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

# SBLAS_SPMM_VARIANT values (kernels.h): every stage-2 selection the library can be pinned to
SPMM_VARIANTS = ["auto", "dpp", "rows", "lanes", "merge", "mfma", "nomfma"]

RTOL = 1e-10          # north_star: "fp64 within 1e-10 rel"
ATOL = 1e-12          # floor for results near zero (inputs are O(1), sums O(1e2))


def close(got, ref):
    return np.allclose(got, ref, rtol=RTOL, atol=ATOL)


class Dev:
    """A CSR matrix resident on the GPU (torch owns the memory; the C ABI gets raw pointers)."""

    def __init__(self, torch, dev, rowptr, colidx, val, cols):
        self.rows = len(rowptr) - 1
        self.cols = cols
        self.h = (np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(colidx, np.int32), np.ascontiguousarray(val, np.float64))
        self.rowptr = torch.from_numpy(self.h[0]).to(dev)
        self.colidx = torch.from_numpy(self.h[1]).to(dev)
        self.val = torch.from_numpy(self.h[2]).to(dev)


def gpu_spmm(sblas, torch, dev, A, B_host, ldb, n, alpha, beta, C_host, ldc):
    B = torch.from_numpy(B_host).to(dev)
    C = torch.from_numpy(C_host.copy()).to(dev)
    nbytes = sblas.spmm_workspace_bytes(A.rows, A.cols, len(A.h[1]), n)
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=dev) if nbytes else None   # 0 bytes: a NULL workspace
    sblas.spmm(A.rows, A.cols, A.rowptr, A.colidx, A.val, B, ldb, n, alpha, beta, C, ldc, ws)
    torch.cuda.synchronize()
    return C.cpu().numpy()


def oracle_spmm_ld(oracle, A, B_host, ldb, n, alpha, beta, C_host, ldc):
    """Oracle works on packed col-major arrays; pack/unpack the leading dimensions around it."""
    rp, ci, v = A.h
    Bp = np.ascontiguousarray(B_host.reshape(-1)[: ldb * n].reshape(n, ldb)[:, : A.cols]).reshape(-1) if A.cols else np.zeros(0)
    Cfull = C_host.copy()
    Cv = Cfull.reshape(-1)[: ldc * n].reshape(n, ldc)
    Cp = np.ascontiguousarray(Cv[:, : A.rows]).reshape(-1)
    oracle.spmm(A.rows, A.cols, n, rp, ci, v, Bp, Cp, alpha, beta)
    Cv[:, : A.rows] = Cp.reshape(n, A.rows)
    return Cfull


def _env_switch(name):
    """Set an SBLAS_* switch for one test; the library caches its switches, so tell it to read them again."""
    import sblas_amd
    old = os.environ.get(name)

    def setter(v):
        os.environ[name] = v
        sblas_amd.reload_env()
    yield setter
    if old is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = old
    sblas_amd.reload_env()


@pytest.fixture
def variant_env():
    yield from _env_switch("SBLAS_SPMM_VARIANT")


@pytest.fixture
def panel_rows_env():
    yield from _env_switch("SBLAS_SPMM_PANEL_ROWS")


@pytest.fixture(scope="module")
def env(sblas, oracle, cuda):
    import torch
    assert sblas.lib().sblas_hip_device_count() >= 1
    return sblas, oracle, torch, cuda


def test_ash85_spmv_known_answers(env, ash85):
    """BASELINE config 2: ash85 SpMV fp64 on 1 GPU vs the CPU verifier (x = y0 = 1)."""
    sblas, oracle, torch, dev = env
    with open(os.path.join(GOLDEN, "ash85_golden.json")) as f:
        gold = json.load(f)
    A = Dev(torch, dev, ash85["rowptr"], ash85["colidx"], ash85["val"], 85)
    for key in ("spmv_a1_b1", "spmv_a3_b4"):
        g = gold[key]
        x = torch.ones(85, dtype=torch.float64, device=dev)
        y = torch.ones(85, dtype=torch.float64, device=dev)
        sblas.spmv(85, 85, A.rowptr, A.colidx, A.val, x, g["alpha"], g["beta"], y)
        got = y.cpu().numpy()
        ref = oracle.spmv(85, *A.h, np.ones(85), np.ones(85), g["alpha"], g["beta"])
        assert close(got, ref)
        assert (got[0], got[1], got[84], got.sum()) == (g["y_first"], g["y_second"], g["y_last"], g["y_sum"])  # small integers: exact


@pytest.mark.parametrize("key", ["spmm_n64_a1_b1", "spmm_n256_a3_b4"])
def test_ash85_spmm_known_answers(env, ash85, key):
    """BASELINE config 1 inputs (and unit_test.cu:183's N=256, alpha=3, beta=4) on the GPU path."""
    sblas, oracle, torch, dev = env
    with open(os.path.join(GOLDEN, "ash85_golden.json")) as f:
        g = json.load(f)[key]
    N = g["N"]
    A = Dev(torch, dev, ash85["rowptr"], ash85["colidx"], ash85["val"], 85)
    B = oracle.rand0to1(85 * N)
    C0 = np.full(85 * N, g["C0"])
    got = gpu_spmm(sblas, torch, dev, A, B, 85, N, g["alpha"], g["beta"], C0, 85)
    ref = oracle.spmm(85, 85, N, *A.h, B, C0.copy(), g["alpha"], g["beta"])
    assert close(got, ref)
    for idx, name in ((0, "C_first"), (1, "C_second"), (-1, "C_last")):
        assert abs(got[idx] - g[name]) <= RTOL * abs(g[name])
    assert abs(got.sum() - g["C_sum"]) <= 1e-9 * g["C_sum"]
    assert oracle.lib().orc_check_equal(ref, got, got.size) == 1          # the reference's own criterion (1e-3 abs)


@pytest.mark.parametrize("variant", SPMM_VARIANTS)
@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 9, 16, 17, 31, 32, 33, 63, 64, 65, 100, 128, 130, 256])
def test_spmm_every_column_count(env, variant_env, variant, n):
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    variant_env(variant)
    rp, ci, v = synth.random_csr(150, 97, 9, seed=n, empty_every=11, long_row=(5, 333))
    A = Dev(torch, dev, rp, ci, v, 97)
    rng = np.random.default_rng(n)
    B = rng.standard_normal(97 * n)
    C0 = rng.standard_normal(150 * n)
    for alpha, beta in ((1.0, 1.0), (-2.5, 0.75), (1.0, 0.0)):
        got = gpu_spmm(sblas, torch, dev, A, B, 97, n, alpha, beta, C0, 150)
        ref = oracle.spmm(150, 97, n, *A.h, B, C0.copy(), alpha, beta)
        assert close(got, ref), (n, alpha, beta, np.abs(got - ref).max())


@pytest.mark.parametrize("n", [9, 16, 17, 31, 32])
def test_spmm_narrow_kernels_still_correct(env, ash85, n):
    """9..32 columns run on the 64-column kernels by default; SBLAS_SPMM_MIN_LDBT=0 selects the 16- / 32-column
    kernels again (kept for A/B runs): both must match the oracle."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows = 700
    rp, ci, v = synth.banded(rows, 40, 90)
    A = Dev(torch, dev, rp, ci, v, rows)
    rng = np.random.default_rng(n)
    B, C0 = rng.standard_normal(rows * n), rng.standard_normal(rows * n)
    ref = oracle.spmm(rows, rows, n, *A.h, B, C0.copy(), 0.5, 2.0)
    old = os.environ.get("SBLAS_SPMM_MIN_LDBT")
    try:
        for setting in ("0", "64"):
            os.environ["SBLAS_SPMM_MIN_LDBT"] = setting
            sblas.reload_env()
            assert close(gpu_spmm(sblas, torch, dev, A, B, rows, n, 0.5, 2.0, C0, rows), ref), (n, setting)
    finally:
        if old is None:
            os.environ.pop("SBLAS_SPMM_MIN_LDBT", None)
        else:
            os.environ["SBLAS_SPMM_MIN_LDBT"] = old
        sblas.reload_env()


@pytest.mark.parametrize("n", [8, 64, 96])
def test_spmm_leading_dimensions_and_untouched_padding(env, n):
    """ldb > K and ldc > M (method 2 writes into Ccopy with ldc = M != m_i, spmm.h:224-231): the rows of C
    between M and ldc must not be touched."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    M, K, ldb, ldc = 77, 53, 60, 90
    rp, ci, v = synth.random_csr(M, K, 6, seed=3)
    A = Dev(torch, dev, rp, ci, v, K)
    rng = np.random.default_rng(0)
    B = rng.standard_normal(ldb * n)
    C0 = rng.standard_normal(ldc * n)
    got = gpu_spmm(sblas, torch, dev, A, B, ldb, n, 1.5, -0.5, C0, ldc)
    ref = oracle_spmm_ld(oracle, A, B, ldb, n, 1.5, -0.5, C0, ldc)
    assert close(got, ref)
    pad_got = got.reshape(n, ldc)[:, M:]
    assert (pad_got == C0.reshape(n, ldc)[:, M:]).all()


def test_spmm_nonfinite_b_rows_not_referenced_stay_out(env, variant_env):
    """Masked DPP slots must not touch real B rows: an Inf/NaN in a B row that no nonzero references must not
    leak into C (0 * Inf = NaN); the reference's CPU loop never reads such rows."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows, n = 300, 64
    rp, ci, v = synth.banded(rows, 7, 40)            # 7 nonzeros per row: every sweep has masked slots
    ci = ci.copy()
    ci[ci == 0] = 1                                  # nobody references B row 0 ...
    used = np.zeros(rows, bool)
    used[ci] = True
    A = Dev(torch, dev, rp, ci, v, rows)
    rng = np.random.default_rng(4)
    B = rng.standard_normal(rows * n).reshape(n, rows)
    B[:, ~used] = np.inf                             # ... nor these rows
    B[:, 0] = np.nan
    B = np.ascontiguousarray(B).reshape(-1)
    C0 = rng.standard_normal(rows * n)
    for variant in SPMM_VARIANTS:
        variant_env(variant)
        got = gpu_spmm(sblas, torch, dev, A, B, rows, n, 1.0, 1.0, C0, rows)
        ref = oracle.spmm(rows, rows, n, *A.h, B, C0.copy(), 1.0, 1.0)
        assert np.isfinite(ref).all() and np.isfinite(got).all() and close(got, ref), variant


def test_spmm_beta_zero_does_not_read_c(env):
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rp, ci, v = synth.random_csr(40, 40, 5, seed=9)
    A = Dev(torch, dev, rp, ci, v, 40)
    B = np.random.default_rng(1).standard_normal(40 * 64)
    C0 = np.full(40 * 64, np.nan)
    got = gpu_spmm(sblas, torch, dev, A, B, 40, 64, 2.0, 0.0, C0, 40)
    ref = oracle.spmm(40, 40, 64, *A.h, B, np.zeros(40 * 64), 2.0, 0.0)
    assert np.isfinite(got).all() and close(got, ref)


@pytest.mark.parametrize("variant", ["auto", "dpp"])
def test_spmm_degenerate_shapes(env, variant_env, variant):
    sblas, oracle, torch, dev = env
    variant_env(variant)
    # nnz = 0: C = beta*C
    rp = np.zeros(11, np.int32)
    A = Dev(torch, dev, rp, np.zeros(0, np.int32), np.zeros(0), 6)
    C0 = np.arange(10 * 64, dtype=np.float64)
    got = gpu_spmm(sblas, torch, dev, A, np.ones(6 * 64), 6, 64, 1.0, 3.0, C0, 10)
    assert (got == 3.0 * C0).all()
    # K = 0 (no columns at all), rows > 0, wide N: C = beta*C with no workspace (this shape used to reach
    # the kernels with a NULL staging copy); beta = 0 must clear C without reading it
    A0 = Dev(torch, dev, np.zeros(301, np.int32), np.zeros(0, np.int32), np.zeros(0), 0)
    for n0 in (1, 8, 64, 130):
        Cn = np.arange(300 * n0, dtype=np.float64) + 1.0
        assert sblas.spmm_workspace_bytes(300, 0, 0, n0) == 0
        got = gpu_spmm(sblas, torch, dev, A0, np.zeros(1), 1, n0, 2.0, -1.5, Cn, 300)
        assert (got == -1.5 * Cn).all(), n0
        got = gpu_spmm(sblas, torch, dev, A0, np.zeros(1), 1, n0, 2.0, 0.0, np.full(300 * n0, np.nan), 300)
        assert (got == 0.0).all(), n0
    # ... and through the split entry point (no staging copy exists for K = 0)
    Cd = torch.from_numpy(np.arange(300 * 64, dtype=np.float64)).to(dev)
    sblas.spmm_rowmajorB(300, 0, A0.rowptr, A0.colidx, A0.val, None, 64, 1.0, 0.5, Cd, 300)
    assert (Cd.cpu().numpy() == 0.5 * np.arange(300 * 64)).all()
    # nnz = 0 with K > 0 and a padded C (ldc > rows): the padding rows stay untouched
    Cp = np.arange(12 * 64, dtype=np.float64)
    A = Dev(torch, dev, rp, np.zeros(0, np.int32), np.zeros(0), 6)
    got = gpu_spmm(sblas, torch, dev, A, np.ones(6 * 64), 6, 64, 1.0, 3.0, Cp, 12).reshape(64, 12)
    want = Cp.copy().reshape(64, 12)
    want[:, :10] *= 3.0
    assert (got == want).all()
    # one row, one column, one nonzero
    A = Dev(torch, dev, np.array([0, 1], np.int32), np.array([0], np.int32), np.array([2.0]), 1)
    got = gpu_spmm(sblas, torch, dev, A, np.array([4.0]), 1, 1, 1.0, 1.0, np.array([1.0]), 1)
    assert got.tolist() == [9.0]
    # a single row much longer than a wavefront, duplicates included
    K = 1000
    ci = (np.arange(5000) * 7 % K).astype(np.int32)
    v = np.random.default_rng(2).standard_normal(5000)
    A = Dev(torch, dev, np.array([0, 5000], np.int32), ci, v, K)
    B = np.random.default_rng(3).standard_normal(K * 64)
    got = gpu_spmm(sblas, torch, dev, A, B, K, 64, 1.0, 0.0, np.zeros(64), 1)
    ref = oracle.spmm(1, K, 64, *A.h, B, np.zeros(64), 1.0, 0.0)
    assert close(got, ref)


def test_spmm_rows_not_multiple_of_panel_and_rectangular(env):
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    for (M, K) in ((1, 300), (31, 5), (33, 64), (257, 129), (1000, 10)):
        rp, ci, v = synth.random_csr(M, K, 12, seed=M)
        A = Dev(torch, dev, rp, ci, v, K)
        rng = np.random.default_rng(M)
        B, C0 = rng.standard_normal(K * 64), rng.standard_normal(M * 64)
        got = gpu_spmm(sblas, torch, dev, A, B, K, 64, 1.0, 1.0, C0, M)
        assert close(got, oracle.spmm(M, K, 64, *A.h, B, C0.copy(), 1.0, 1.0)), (M, K)


def test_spmm_nd24k_like_reduced(env):
    """The bench workload's structure (399 nnz/row, band +-2000) at 3 % of its rows, vs the oracle."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows, (rp, ci, v) = synth.nd24k_like(scale=0.03)
    A = Dev(torch, dev, rp, ci, v, rows)
    rng = np.random.default_rng(7)
    B, C0 = rng.random(rows * 64), np.ones(rows * 64)
    got = gpu_spmm(sblas, torch, dev, A, B, rows, 64, 1.0, 1.0, C0, rows)
    ref = oracle.spmm(rows, rows, 64, *A.h, B, C0.copy(), 1.0, 1.0)
    assert close(got, ref)


@pytest.mark.parametrize("g", [1, 2, 4, 8])
def test_method1_column_blocks(env, ash85, g):
    """Method 1 (spmm.h:83-161): GPU i owns columns [i*ceil(N/g), ...) of B and C with the full A; the blocks
    together must equal the verifier.  (All blocks run on this one GPU; placement is host arithmetic.)"""
    sblas, oracle, torch, dev = env
    N = 64
    A = Dev(torch, dev, ash85["rowptr"], ash85["colidx"], ash85["val"], 85)
    B = oracle.rand0to1(85 * N)
    C = np.ones(85 * N)
    for i in range(g):
        off, dim = sblas.partition_dense(N, g, i)
        blk = gpu_spmm(sblas, torch, dev, A, B[off * 85:(off + dim) * 85].copy(), 85, dim, 3.0, 4.0, C[off * 85:(off + dim) * 85].copy(), 85)
        C[off * 85:(off + dim) * 85] = blk
    ref = oracle.spmm(85, 85, N, *A.h, B, np.ones(85 * N), 3.0, 4.0)
    assert close(C, ref)


@pytest.mark.parametrize("g", [1, 2, 4, 8])
def test_method2_row_blocks_merge_and_epilogue(env, ash85, g):
    """Method 2 (spmm.h:163-284): nnz-balanced row blocks with re-based row pointers write A_i*B (alpha=beta=1)
    into a zeroed Ccopy at row offset start_row with ldc = M; the partials are summed (the all-reduce's
    arithmetic) and C = beta*C + alpha*Ccopy (kernel.h:27-38) finishes.  Boundary rows are split between two
    blocks, so only <=1e-10 relative is possible, not bit equality."""
    sblas, oracle, torch, dev = env
    N, M = 64, 85
    rp, ci, v = ash85["rowptr"], ash85["colidx"], ash85["val"]
    Bh = oracle.rand0to1(M * N)
    B = torch.from_numpy(Bh).to(dev)
    ws = torch.empty(sblas.spmm_workspace_bytes(M, M, 523, N) // 8, dtype=torch.float64, device=dev)
    total = torch.zeros(M * N, dtype=torch.float64, device=dev)
    for i in range(g):
        d = sblas.partition_nnz(rp, g, i)
        lo, k = d["first_nnz"], d["nnz"]
        Ai = Dev(torch, dev, d["rowptr"], ci[lo:lo + k], v[lo:lo + k], M)
        ccopy = torch.zeros(M * N, dtype=torch.float64, device=dev)
        sblas.spmm(Ai.rows, M, Ai.rowptr, Ai.colidx, Ai.val, B, M, N, 1.0, 1.0, ccopy, M, ws, c_offset=d["start_row"])
        sblas.axpby(M * N, 1.0, ccopy, 1.0, total)          # rank-order sum = what the all-reduce computes
    C = torch.ones(M * N, dtype=torch.float64, device=dev)
    sblas.axpby(M * N, 3.0, total, 4.0, C)
    ref = oracle.spmm(M, M, N, rp, ci, v, Bh, np.ones(M * N), 3.0, 4.0)
    assert close(C.cpu().numpy(), ref)


@pytest.mark.parametrize("g", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("case", ["ash85", "long_row"])
def test_method2_rowblock_merge_fast_path(env, ash85, g, case):
    """the same method-2 result without the M x N zero fill, the all-reduce and the axpby pass: every
    block computes its own rows into a packed buffer (beta = 0, ldc = rows of the block) and
    sblas_hip_merge_rowblocks_local_f64 scatters the blocks and applies alpha / beta.  `long_row`: one row holds
    half of the nonzeros, so it is cut into pieces on several blocks (more than two terms for that row)."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    if case == "ash85":
        M = K = 85
        rp, ci, v = ash85["rowptr"], ash85["colidx"], ash85["val"]
    else:
        M, K = 300, 2000
        rp, ci, v = synth.random_csr(M, K, 6, seed=5, long_row=(150, 1800), sorted_rows=True)
    N = 24
    rng = np.random.default_rng(3)
    Bh, C0 = rng.standard_normal(K * N), rng.standard_normal(M * N)
    B = torch.from_numpy(Bh).to(dev)
    blocks, starts, nrows = [], [], []
    for i in range(g):
        d = sblas.partition_nnz(rp, g, i)
        lo, k = d["first_nnz"], d["nnz"]
        m_i = len(d["rowptr"]) - 1
        starts.append(d["start_row"])
        nrows.append(m_i)
        blk = torch.full((max(m_i * N, 1),), 7.0, dtype=torch.float64, device=dev)   # junk: beta = 0 must not read it
        if m_i > 0:
            Ai = Dev(torch, dev, d["rowptr"], ci[lo:lo + k], v[lo:lo + k], K)
            ws = torch.empty(sblas.spmm_workspace_bytes(m_i, K, k, N) // 8, dtype=torch.float64, device=dev)
            sblas.spmm(m_i, K, Ai.rowptr, Ai.colidx, Ai.val, B, K, N, 1.0, 0.0, blk, m_i, ws)
        blocks.append(blk)
    C = torch.from_numpy(C0.copy()).to(dev)
    sblas.merge_rowblocks_local(M, N, starts, nrows, blocks, 3.0, 4.0, C)
    ref = oracle.spmm(M, K, N, rp, ci, v, Bh, C0.copy(), 3.0, 4.0)
    assert close(C.cpu().numpy(), ref)
    # N = 1 (the SpMV merge) with beta = 0
    xh = rng.standard_normal(K)
    x = torch.from_numpy(xh).to(dev)
    parts = []
    for i in range(g):
        d = sblas.partition_nnz(rp, g, i)
        lo, k = d["first_nnz"], d["nnz"]
        m_i = len(d["rowptr"]) - 1
        yb = torch.zeros(max(m_i, 1), dtype=torch.float64, device=dev)
        if m_i > 0:
            Ai = Dev(torch, dev, d["rowptr"], ci[lo:lo + k], v[lo:lo + k], K)
            sblas.spmv(m_i, K, Ai.rowptr, Ai.colidx, Ai.val, x, 1.0, 0.0, yb)
        parts.append(yb)
    y = torch.full((M,), float("nan"), dtype=torch.float64, device=dev)
    sblas.merge_rowblocks_local(M, 1, starts, nrows, parts, 2.0, 0.0, y)
    assert close(y.cpu().numpy(), oracle.spmv(M, rp, ci, v, xh, np.zeros(M), 2.0, 0.0))


def _config4_blocks(sblas, torch, dev_of, rp, ci, v, K, g):
    """The nnz row-block partition of CsrSparseMatrix::sync2gpu(segment) (matrix.h:356-395), block q on dev_of(q)."""
    blocks = []
    for q in range(g):
        d = sblas.partition_nnz(rp, g, q)
        lo, k = d["first_nnz"], d["nnz"]
        blocks.append(dict(start=d["start_row"], m=len(d["rowptr"]) - 1, nnz=k,
                           A=Dev(torch, dev_of(q), d["rowptr"], ci[lo:lo + k], v[lo:lo + k], K)))
    return blocks


def _config4_run(sblas, oracle, torch, devs, rows, rp, ci, v, N, merge, check_rows=None):
    """BASELINE config 4: method 2, g = len(devs) nnz row blocks, N dense columns, both merges of spmm.h's v2.
    devs all equal: ranks folded onto one device (how the 1-GPU box rehearses it); distinct: RCCL over xGMI."""
    g, K, M = len(devs), rows, rows
    alpha, beta = 3.0, 4.0
    tdev = [torch.device("cuda", d) for d in devs]
    Bh = oracle.rand0to1(K * N)                                   # the reference's B (matrix.h:519-528)
    C0 = np.ones(M * N)
    blocks = _config4_blocks(sblas, torch, lambda q: tdev[q], rp, ci, v, K, g)
    comm = sblas.comm_get(devs)
    Bs, Cs, streams, part, gath, ws = [], [], [], [], [], []
    total_blocks = sum(b["m"] for b in blocks) * N
    for q in range(g):
        with torch.cuda.device(tdev[q]):
            Bs.append(torch.from_numpy(Bh).to(tdev[q]))
            Cs.append(torch.from_numpy(C0.copy()).to(tdev[q]))
            streams.append(torch.cuda.Stream(device=tdev[q]))
            b = blocks[q]
            ws.append(torch.empty(max(sblas.spmm_workspace_bytes(b["m"], K, b["nnz"], N) // 8, 1), dtype=torch.float64, device=tdev[q]))
            if merge == "allreduce":                              # spmm.h:182-183, 248-251: zeroed M x N, ldc = M
                part.append(torch.zeros(M * N, dtype=torch.float64, device=tdev[q]))
            else:                                                 # packed m_q x N, beta = 0
                part.append(torch.full((max(b["m"] * N, 1),), 7.0, dtype=torch.float64, device=tdev[q]))
                gath.append(torch.empty(max(total_blocks, 1), dtype=torch.float64, device=tdev[q]))
    torch.cuda.synchronize()
    for q in range(g):
        b = blocks[q]
        with torch.cuda.device(tdev[q]):
            if b["m"] == 0:
                continue
            if merge == "allreduce":
                sblas.spmm(b["m"], K, b["A"].rowptr, b["A"].colidx, b["A"].val, Bs[q], K, N, 1.0, 1.0, part[q], M, ws[q],
                           stream=streams[q], c_offset=b["start"])
            else:
                sblas.spmm(b["m"], K, b["A"].rowptr, b["A"].colidx, b["A"].val, Bs[q], K, N, 1.0, 0.0, part[q], b["m"], ws[q],
                           stream=streams[q])
    if merge == "allreduce":
        sblas.allreduce_sum(comm, part, streams, M * N)           # spmm.h:260-262
        for q in range(g):
            with torch.cuda.device(tdev[q]):
                sblas.axpby(M * N, alpha, part[q], beta, Cs[q], stream=streams[q])   # spmm.h:283
    else:
        sblas.merge_rowblocks(comm, M, N, [b["start"] for b in blocks], [b["m"] for b in blocks], part, gath, alpha, beta,
                              Cs, M, streams)
    for d in set(devs):
        torch.cuda.synchronize(d)
    if check_rows is None:
        ref = oracle.spmm(M, K, N, rp, ci, v, Bh, C0.copy(), alpha, beta)
        for q in range(g):
            got = Cs[q].cpu().numpy()
            assert close(got, ref), (merge, q, np.abs(got - ref).max())
    else:                                                         # full size: row windows against the oracle
        ref = C0.copy()
        for r0 in check_rows:
            oracle.spmm_rows(r0, r0 + 64, M, K, N, rp, ci, v, Bh, ref, alpha, beta)
        for q in range(g):
            got = Cs[q].view(N, M).cpu().numpy()
            for r0 in check_rows:
                assert close(got[:, r0:r0 + 64], ref.reshape(N, M)[:, r0:r0 + 64]), (merge, q, r0)
        # every rank holds the same C (bit for bit with the row-block merge: same terms in the same order)
        for q in range(1, g):
            if merge != "allreduce":
                assert torch.equal(Cs[q].cpu(), Cs[0].cpu())
    return Cs[0]


@pytest.mark.parametrize("merge", ["allreduce", "rowblocks"])
def test_config4_method2_n128_g4(env, merge):
    """BASELINE config 4 at reduced rows (nd24k-like, 399 per row; the oracle finishes in seconds): method 2, g = 4 nnz
    row blocks, N = 128, the reference's call pattern (spmm.h:199-251: ldc = M != m_i, C view at start_row) with the
    all-reduce merge, and the packed beta = 0 form with the row-block merge -- every rank's full C against the oracle."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows, (rp, ci, v) = synth.nd24k_like(scale=0.05)
    _config4_run(sblas, oracle, torch, [dev.index or 0] * 4, rows, rp, ci, v, 128, merge)


@pytest.mark.parametrize("merge", ["allreduce", "rowblocks"])
def test_config4_method2_full_size(env, merge):
    """The same at BASELINE's full 72 000 rows: row windows at both ends, at a block boundary and in the middle against
    the oracle, and the ranks' results identical."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows, (rp, ci, v) = synth.nd24k_like()
    cut = sblas.partition_nnz(rp, 4, 1)["start_row"]
    _config4_run(sblas, oracle, torch, [dev.index or 0] * 4, rows, rp, ci, v, 128, merge,
                 check_rows=[0, max(cut - 32, 0), rows // 2, rows - 64])


@pytest.mark.parametrize("merge", ["allreduce", "rowblocks"])
@pytest.mark.parametrize("g", [2, 4])
def test_method2_over_real_devices(env, merge, g):
    """comm.hip's multi-device branches (ncclCommInitAll, grouped ncclAllReduce, grouped ncclSend / ncclRecv) over
    DISTINCT devices -- runs wherever the process sees at least g GPUs (the round-end 8-GPU node); on the one-GPU box
    the same code runs folded (the tests above)."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    if torch.cuda.device_count() < g:
        pytest.skip("needs %d GPUs, this box has %d (the folded variant above ran instead)" % (g, torch.cuda.device_count()))
    rows, (rp, ci, v) = synth.nd24k_like(scale=0.05)
    _config4_run(sblas, oracle, torch, list(range(g)), rows, rp, ci, v, 128, merge)
    # SpMV merge (N = 1) over the same communicator
    comm = sblas.comm_get(list(range(g)))
    xh = np.random.default_rng(2).standard_normal(rows)
    yh = np.random.default_rng(3).standard_normal(rows)
    parts, ys, streams, starts, nrows = [], [], [], [], []
    for q in range(g):
        d = sblas.partition_nnz(rp, g, q)
        lo, k = d["first_nnz"], d["nnz"]
        td = torch.device("cuda", q)
        with torch.cuda.device(td):
            Ai = Dev(torch, td, d["rowptr"], ci[lo:lo + k], v[lo:lo + k], rows)
            m_i = len(d["rowptr"]) - 1
            st = torch.cuda.Stream(device=td)
            yb = torch.zeros(max(m_i, 1), dtype=torch.float64, device=td)
            x = torch.from_numpy(xh).to(td)
            torch.cuda.synchronize(td)
            sblas.spmv(m_i, rows, Ai.rowptr, Ai.colidx, Ai.val, x, 1.0, 0.0, yb, stream=st)
            parts.append(yb); streams.append(st); starts.append(d["start_row"]); nrows.append(m_i)
            ys.append(torch.from_numpy(yh.copy()).to(td))
    gath = [torch.empty(max(sum(nrows), 1), dtype=torch.float64, device=torch.device("cuda", q)) for q in range(g)]
    sblas.merge_rowblocks(comm, rows, 1, starts, nrows, parts, gath, 2.0, -1.0, ys, rows, streams)
    for q in range(g):
        torch.cuda.synchronize(q)
    ref = oracle.spmv(rows, rp, ci, v, xh, yh.copy(), 2.0, -1.0)
    for q in range(g):
        assert close(ys[q].cpu().numpy(), ref), q


@pytest.mark.parametrize("avg", [1, 3, 7, 15, 30, 70, 400])
def test_spmv_every_row_length_class(env, avg):
    """One case per lanes-per-row instantiation (4..64), unsorted rows, empty rows, a long row."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    M, K = 1234, 987
    rp, ci, v = synth.random_csr(M, K, avg, seed=avg, empty_every=13, long_row=(17, 2000))
    A = Dev(torch, dev, rp, ci, v, K)
    rng = np.random.default_rng(avg)
    xh, yh = rng.standard_normal(K), rng.standard_normal(M)
    for alpha, beta in ((1.0, 1.0), (0.5, -2.0), (3.0, 0.0)):
        x, y = torch.from_numpy(xh).to(dev), torch.from_numpy(yh.copy()).to(dev)
        sblas.spmv(M, K, A.rowptr, A.colidx, A.val, x, alpha, beta, y)
        ref = oracle.spmv(M, *A.h, xh, yh.copy(), alpha, beta)
        assert close(y.cpu().numpy(), ref), (avg, alpha, beta)


def test_spmv_row_block_offsets(env, ash85):
    """spmv.h:85-88: y view = Ccopy + starting_row; same partition + merge as method 2, N = 1."""
    sblas, oracle, torch, dev = env
    rp, ci, v = ash85["rowptr"], ash85["colidx"], ash85["val"]
    x = torch.ones(85, dtype=torch.float64, device=dev)
    for g in (2, 4, 8):
        total = torch.zeros(85, dtype=torch.float64, device=dev)
        for i in range(g):
            d = sblas.partition_nnz(rp, g, i)
            lo, k = d["first_nnz"], d["nnz"]
            Ai = Dev(torch, dev, d["rowptr"], ci[lo:lo + k], v[lo:lo + k], 85)
            sblas.spmv(Ai.rows, 85, Ai.rowptr, Ai.colidx, Ai.val, x, 1.0, 1.0, total, y_offset=d["start_row"])
        y = torch.ones(85, dtype=torch.float64, device=dev)
        sblas.axpby(85, 3.0, total, 4.0, y)
        ref = oracle.spmv(85, rp, ci, v, np.ones(85), np.ones(85), 3.0, 4.0)
        assert close(y.cpu().numpy(), ref)


@pytest.mark.parametrize("n", [1, 2, 3, 255, 256, 257, 100001])
def test_axpby(env, n):
    sblas, oracle, torch, dev = env
    rng = np.random.default_rng(n)
    xh, yh = rng.standard_normal(n + 1), rng.standard_normal(n + 1)
    for off in (0, 1):                                  # off=1: 8-byte-aligned only -> scalar path
        x = torch.from_numpy(xh).to(dev)[off:off + n].contiguous() if off == 0 else torch.from_numpy(xh).to(dev)[1:]
        y = torch.from_numpy(yh.copy()).to(dev)[off:off + n] if off else torch.from_numpy(yh[:n].copy()).to(dev)
        x = x[:n]
        sblas.axpby(n, 0.3, x, -1.7, y)
        ref = yh[off:off + n].copy()
        oracle.lib().orc_axpby(n, 0.3, np.ascontiguousarray(xh[off:off + n] if off else xh[:n]), -1.7, ref)
        assert close(y.cpu().numpy(), ref)


def test_full_size_properties(env):
    """BASELINE config 3 size (72 000 rows, 28.7 M nnz, N = 64), where the oracle would take minutes:
    size-independent properties instead -- (i) B = ones gives C[i, :] = rowsum(A)[i]; (ii) column j of SpMM
    equals SpMV with B[:, j]; (iii) linearity in B; plus (iv) an oracle spot check on 192 sampled rows."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows, (rp, ci, v) = synth.nd24k_like()
    N = 64
    A = Dev(torch, dev, rp, ci, v, rows)
    ws = torch.empty(sblas.spmm_workspace_bytes(rows, rows, len(ci), N) // 8, dtype=torch.float64, device=dev)

    def run(Bt_, alpha=1.0, beta=0.0, C=None):
        C = torch.zeros(rows * N, dtype=torch.float64, device=dev) if C is None else C
        sblas.spmm(rows, rows, A.rowptr, A.colidx, A.val, Bt_, rows, N, alpha, beta, C, rows, ws)
        return C

    # (i) row sums, computed independently with torch on the GPU
    ones = torch.ones(rows * N, dtype=torch.float64, device=dev)
    C1 = run(ones).view(N, rows)
    lens = torch.from_numpy(np.diff(rp).astype(np.int64)).to(dev)
    rowsum = torch.segment_reduce(A.val, "sum", lengths=lens)
    assert torch.allclose(C1, rowsum.expand(N, rows), rtol=1e-10, atol=1e-12)
    # (ii) SpMM column == SpMV
    g = torch.Generator(device="cpu").manual_seed(211)
    B = torch.rand(rows * N, dtype=torch.float64, generator=g).to(dev)
    CB = run(B)
    for j in (0, 17, 63):
        y = torch.zeros(rows, dtype=torch.float64, device=dev)
        sblas.spmv(rows, rows, A.rowptr, A.colidx, A.val, B[j * rows:(j + 1) * rows].contiguous(), 1.0, 0.0, y)
        assert torch.allclose(CB.view(N, rows)[j], y, rtol=1e-10, atol=1e-12)
    # (iii) linearity: A(2B + ones) = 2AB + A*ones
    Clin = run(2.0 * B + ones)
    assert torch.allclose(Clin, 2.0 * CB + C1.reshape(-1), rtol=1e-10, atol=1e-11)
    # (iv) oracle on sampled rows (alpha, beta != trivial), same inputs
    C0 = torch.full((rows * N,), 1.0, dtype=torch.float64, device=dev)
    got = run(B, 3.0, 4.0, C0).cpu().numpy().reshape(N, rows)
    Bh = B.cpu().numpy()
    sample = np.r_[0:64, rows // 2:rows // 2 + 64, rows - 64:rows]
    ref = np.ones(rows * N)
    for r0 in (0, rows // 2, rows - 64):
        oracle.spmm_rows(r0, r0 + 64, rows, rows, N, rp, ci, v, Bh, ref, 3.0, 4.0)
    assert close(got[:, sample], ref.reshape(N, rows)[:, sample])


def test_config5_full_size_on_one_gpu(env):
    """BASELINE config 5's size on one GPU: 4 147 110 rows (Queen_4147's), grid-structured Queen-like rows (~275 M
    nonzeros), N = 256.  Bt for 256 columns would be 8.5 GB, beyond the kernels' 32-bit byte offsets, so the call walks
    two 128-column chunks over a 4.25 GB staging copy (row offsets up to 99 % of 2^32) -- with the row-merging direct
    kernel.  Properties: B = ones gives the row sums in every column; oracle windows at both ends and in the middle
    on a random B with alpha, beta != trivial."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rp, ci, v = synth.queen_like_grid(4147110)
    rows = len(rp) - 1
    N = 256
    assert rows > 4_000_000 and (rows + 1) * 256 * 8 > 2 ** 32
    A = Dev(torch, dev, rp, ci, v, rows)
    ws = torch.empty(sblas.spmm_workspace_bytes(rows, rows, len(ci), N) // 8, dtype=torch.float64, device=dev)
    assert ws.numel() * 8 < 5 * 2 ** 30                            # one 128-column chunk, not 256 columns
    C = torch.zeros(rows * N, dtype=torch.float64, device=dev)
    B = torch.ones(rows * N, dtype=torch.float64, device=dev)
    sblas.panel_census()
    sblas.spmm(rows, rows, A.rowptr, A.colidx, A.val, B, rows, N, 1.0, 0.0, C, rows, ws)
    census = sblas.panel_census()
    assert census["direct"] > 0 and census["windowed"] == 0, census
    lens = torch.from_numpy(np.diff(rp).astype(np.int64)).to(dev)
    rowsum = torch.segment_reduce(A.val, "sum", lengths=lens)
    for j in (0, 127, 128, 255):                                   # both chunks
        assert torch.allclose(C.view(N, rows)[j], rowsum, rtol=1e-10, atol=1e-11), j
    del lens, rowsum
    g = torch.Generator(device=dev).manual_seed(211)
    B.uniform_(0.0, 1.0, generator=g)
    C.fill_(1.0)
    sblas.spmm(rows, rows, A.rowptr, A.colidx, A.val, B, rows, N, 3.0, 4.0, C, rows, ws)
    Bh = B.cpu().numpy()
    ref = np.ones(rows * N)
    got = C.view(N, rows)
    for r0 in (0, rows // 2, rows - 64):
        oracle.spmm_rows(r0, r0 + 64, rows, rows, N, rp, ci, v, Bh, ref, 3.0, 4.0)
        assert close(got[:, r0:r0 + 64].cpu().numpy(), ref.reshape(N, rows)[:, r0:r0 + 64]), r0
    # One rank's share of method 2 at g = 8 (row block 5, split by nonzeros): only the block's column range of B is
    # staged (15 % of it) -- the workspace is poisoned first, the rows outside the range must not reach C.
    nnz = len(ci)
    a, b = (int(x) for x in np.searchsorted(rp, [nnz * 5 // 8, nnz * 6 // 8]))
    m = b - a
    sub = torch.from_numpy((rp[a:b + 1] - rp[a]).astype(np.int32)).to(dev)
    ws.fill_(float("nan"))
    Cb = torch.ones(m * N, dtype=torch.float64, device=dev)
    sblas.spmm(m, rows, sub, A.colidx[rp[a]:rp[b]], A.val[rp[a]:rp[b]], B, rows, N, 3.0, 4.0, Cb, m, ws)
    torch.cuda.synchronize()
    ldbt = 128
    Bt = ws[: (rows + 1) * ldbt].view(rows + 1, ldbt)
    lo, hi = int(ci[rp[a]:rp[b]].min()), int(ci[rp[a]:rp[b]].max())
    assert hi - lo < 0.25 * rows
    assert bool(torch.isnan(Bt[: lo - 32]).all()) and bool(torch.isnan(Bt[hi + 33: rows - 32]).all())    # never staged
    for r0 in (a, b - 64):
        assert torch.equal(Cb.view(N, m)[:, r0 - a:r0 - a + 64], got[:, r0:r0 + 64]), r0   # same kernels, same sums as the full call


# ---------------------------------------------------------------------------------------------------------
# the windowed (row panel x LDS B tile) kernel and its per-panel fallback
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", SPMM_VARIANTS)
@pytest.mark.parametrize("shape", [(1000, 40, 100, 64), (777, 60, 300, 130), (200, 30, 20, 64), (90, 80, 45, 256)])
def test_spmm_windowed_variants_banded(env, variant_env, variant, shape):
    """Banded, ascending rows: dense enough over their span that the windowed path is taken.  Covers several
    tiles per panel, K not a multiple of the 128-row tile, K < one tile, N > 64 (column tiles) and padding."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows, per_row, half, n = shape
    variant_env(variant)
    rp, ci, v = synth.banded(rows, per_row, half)
    A = Dev(torch, dev, rp, ci, v, rows)
    rng = np.random.default_rng(rows)
    B, C0 = rng.standard_normal(rows * n), rng.standard_normal(rows * n)
    got = gpu_spmm(sblas, torch, dev, A, B, rows, n, 1.25, -0.5, C0, rows)
    ref = oracle.spmm(rows, rows, n, *A.h, B, C0.copy(), 1.25, -0.5)
    assert close(got, ref), (variant, shape, np.abs(got - ref).max())


@pytest.mark.parametrize("variant", ["auto", "mfma", "nomfma"])
@pytest.mark.parametrize("damage", ["all_descending", "one_row_shuffled", "first_col_not_min", "duplicates"])
def test_spmm_windowed_fallback_on_unsorted_rows(env, variant_env, variant, damage):
    """The windowed path expects ascending columns but must never depend on it: panels whose rows break the
    expectation are detected in-kernel and recomputed by the direct loop before C is written."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    variant_env(variant)
    rows, n = 600, 64
    rp, ci, v = synth.banded(rows, 50, 120)
    ci, v = ci.copy(), v.copy()
    rng = np.random.default_rng(11)
    if damage == "all_descending":
        for r in range(rows):
            ci[rp[r]:rp[r + 1]] = ci[rp[r]:rp[r + 1]][::-1]
            v[rp[r]:rp[r + 1]] = v[rp[r]:rp[r + 1]][::-1]
    elif damage == "one_row_shuffled":
        for r in (5, 300, 599):
            perm = rng.permutation(rp[r + 1] - rp[r])
            ci[rp[r]:rp[r + 1]] = ci[rp[r]:rp[r + 1]][perm]
            v[rp[r]:rp[r + 1]] = v[rp[r]:rp[r + 1]][perm]
    elif damage == "first_col_not_min":
        for r in range(0, rows, 7):                 # swap first and last entry: span estimate is wrong
            a, b = rp[r], rp[r + 1] - 1
            ci[a], ci[b] = ci[b], ci[a]
            v[a], v[b] = v[b], v[a]
    else:                                           # repeated column indices inside ascending rows
        for r in range(0, rows, 3):
            ci[rp[r] + 1] = ci[rp[r]]
            ci[rp[r + 1] - 1] = ci[rp[r + 1] - 2]
    A = Dev(torch, dev, rp, ci, v, rows)
    B, C0 = rng.standard_normal(rows * n), rng.standard_normal(rows * n)
    got = gpu_spmm(sblas, torch, dev, A, B, rows, n, 1.0, 1.0, C0, rows)
    ref = oracle.spmm(rows, rows, n, *A.h, B, C0.copy(), 1.0, 1.0)
    assert close(got, ref), (variant, damage, np.abs(got - ref).max())


@pytest.mark.parametrize("variant", ["auto", "mfma", "nomfma", "merge"])
def test_spmm_windowed_mixed_panels_and_row_blocks(env, variant_env, variant):
    """Dense-band panels next to sparse wide-span panels (direct path chosen per panel), empty rows, rows longer
    than several chunks, and a method-2 style row block (re-based row pointers, C offset, ldc > rows)."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    variant_env(variant)
    rows, K, n = 512, 3000, 64
    rng = np.random.default_rng(21)
    lens = np.where(np.arange(rows) % 128 < 64, 200, 3).astype(np.int64)      # alternating dense / sparse panels
    lens[10] = 0
    lens[70] = 0
    lens[200] = 900
    rp = np.zeros(rows + 1, np.int64)
    rp[1:] = np.cumsum(lens)
    ci = np.empty(rp[-1], np.int32)
    for r in range(rows):
        if lens[r] >= 100:
            lo = min(r * 4, K - 1000)
            ci[rp[r]:rp[r + 1]] = np.sort(rng.choice(np.arange(lo, lo + 1000), lens[r], replace=False))
        else:
            ci[rp[r]:rp[r + 1]] = np.sort(rng.choice(K, lens[r], replace=False))
    v = rng.standard_normal(rp[-1])
    A = Dev(torch, dev, rp.astype(np.int32), ci, v, K)
    B, C0 = rng.standard_normal(K * n), rng.standard_normal(rows * n)
    got = gpu_spmm(sblas, torch, dev, A, B, K, n, 2.0, 0.5, C0, rows)
    ref = oracle.spmm(rows, K, n, *A.h, B, C0.copy(), 2.0, 0.5)
    assert close(got, ref)
    # row block [100, 400) written at offset into a taller C
    sub = rp[100:401] - rp[100]
    As = Dev(torch, dev, sub.astype(np.int32), ci[rp[100]:rp[400]], v[rp[100]:rp[400]], K)
    Cbig = rng.standard_normal(rows * n)
    Bd = torch.from_numpy(B).to(dev)
    Cd = torch.from_numpy(Cbig.copy()).to(dev)
    ws = torch.empty(sblas.spmm_workspace_bytes(300, K, 1, n) // 8, dtype=torch.float64, device=dev)
    sblas.spmm(300, K, As.rowptr, As.colidx, As.val, Bd, K, n, 1.0, 1.0, Cd, rows, ws, c_offset=100)
    want = Cbig.copy().reshape(n, rows)
    part = np.zeros(300 * n)
    oracle.spmm(300, K, n, *As.h, B, part, 1.0, 0.0)
    want[:, 100:400] += part.reshape(n, 300)
    assert close(Cd.cpu().numpy(), want.reshape(-1))


@pytest.fixture
def stage_range_env():
    yield from _env_switch("SBLAS_STAGE_RANGE")


@pytest.mark.parametrize("variant", ["auto", "dpp", "mfma", "merge"])
@pytest.mark.parametrize("n", [64, 128, 200])
@pytest.mark.parametrize("kind", ["banded", "unsorted", "scattered"])
def test_spmm_row_block_stages_only_the_rows_of_b_it_refers_to(env, variant_env, stage_range_env, variant, n, kind):
    """A method-2 row block copies only its column range of B to row-major form (chosen by itself when the block is
    large enough for the saved traffic to matter: config 5's eighths; switched on here).  The workspace is
    poisoned with NaN first and B holds Inf outside the range: neither may reach C, the rows of the copy outside the
    range must still hold the poison (they were not staged), and unsorted rows must not shrink the range."""
    sblas, oracle, torch, dev = env
    variant_env(variant)
    stage_range_env("1")
    K, r0, rows, band = 20000, 9000, 1500, 300
    rng = np.random.default_rng(77)
    lens = rng.integers(20, 90, rows)
    lens[5] = 0
    rp = np.zeros(rows + 1, np.int64)
    rp[1:] = np.cumsum(lens)
    ci = np.empty(rp[-1], np.int32)
    for r in range(rows):
        lo, hi = (0, K) if kind == "scattered" else (r0 + r - band, r0 + r + band)
        c = rng.choice(np.arange(lo, hi), lens[r], replace=False)
        ci[rp[r]:rp[r + 1]] = c if kind == "unsorted" else np.sort(c)
    v = rng.standard_normal(rp[-1])
    A = Dev(torch, dev, rp.astype(np.int32), ci, v, K)
    B = rng.standard_normal(K * n)
    Bm = B.reshape(n, K)
    cmin, cmax = int(ci.min()), int(ci.max())
    if kind != "scattered":
        Bm[:, : cmin] = np.inf
        Bm[:, cmax + 1:] = -np.inf
    C0 = rng.standard_normal(rows * n)
    Bd, Cd = torch.from_numpy(B).to(dev), torch.from_numpy(C0.copy()).to(dev)
    ws = torch.full((sblas.spmm_workspace_bytes(rows, K, len(ci), n) // 8,), float("nan"), dtype=torch.float64, device=dev)
    sblas.spmm(rows, K, A.rowptr, A.colidx, A.val, Bd, K, n, 1.5, -0.5, Cd, rows, ws)
    torch.cuda.synchronize()
    Bo = B.copy().reshape(n, K)
    if kind != "scattered":                       # the oracle multiplies every entry it meets: give it finite rows
        Bo[:, : cmin] = 0.0
        Bo[:, cmax + 1:] = 0.0
    ref = oracle.spmm(rows, K, n, *A.h, Bo.reshape(-1), C0.copy(), 1.5, -0.5)
    assert close(Cd.cpu().numpy(), ref)
    if kind != "scattered":
        ldbt = int(sblas.lib().sblas_hip_spmm_ldbt(n))
        Bt = ws[: (K + 1) * ldbt].view(K + 1, ldbt).cpu().numpy()
        assert np.isnan(Bt[: cmin - 32]).all() and np.isnan(Bt[cmax + 33: K - 32]).all()   # never staged
        assert np.array_equal(Bt[cmin: cmax + 1, :n], Bo[:, cmin: cmax + 1].T)              # the range, exactly
        assert not Bt[K].any()                                                              # the all-zero row


@pytest.mark.parametrize("mode", ["0", "1"])
def test_spmm_stage_range_switch_on_square_and_block(env, stage_range_env, mode):
    """SBLAS_STAGE_RANGE=1 sends a full square matrix through the range staging (range = everything), =0 keeps a row
    block on the whole-B staging: same results either way."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    stage_range_env(mode)
    rng = np.random.default_rng(5)
    rp, ci, v = synth.banded(3000, 60, 200, seed=3)
    A = Dev(torch, dev, rp, ci, v, 3000)
    for n in (64, 130):
        B, C0 = rng.standard_normal(3000 * n), rng.standard_normal(3000 * n)
        got = gpu_spmm(sblas, torch, dev, A, B, 3000, n, 2.0, 0.25, C0, 3000)
        assert close(got, oracle.spmm(3000, 3000, n, *A.h, B, C0.copy(), 2.0, 0.25))
        sub = (rp[1000:1501] - rp[1000]).astype(np.int32)
        As = Dev(torch, dev, sub, ci[rp[1000]:rp[1500]], v[rp[1000]:rp[1500]], 3000)
        Cs = rng.standard_normal(500 * n)
        got = gpu_spmm(sblas, torch, dev, As, B, 3000, n, 1.0, 1.0, Cs, 500)
        assert close(got, oracle.spmm(500, 3000, n, *As.h, B, Cs.copy(), 1.0, 1.0))


def test_spmm_panel_census_paths_are_really_taken(env, variant_env, panel_rows_env):
    """The parity tests above cannot tell a windowed run from a fallback run; the census can.  (SBLAS_SPMM_PANEL_ROWS
    pins 48-row panels; [0] windowed, [1] direct kernel, [2] windowed but recomputed by the in-kernel fallback.)"""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    variant_env("nomfma")
    panel_rows_env("48,2")
    n = 64

    def run(rp, ci, v, rows, cols):
        A = Dev(torch, dev, rp, ci, v, cols)
        rng = np.random.default_rng(1)
        B, C0 = rng.standard_normal(cols * n), rng.standard_normal(rows * n)
        sblas.panel_stats()
        got = gpu_spmm(sblas, torch, dev, A, B, cols, n, 1.0, 1.0, C0, rows)
        st = sblas.panel_stats()
        assert close(got, oracle.spmm(rows, cols, n, *A.h, B, C0.copy(), 1.0, 1.0))
        return st

    rows = 480
    rp, ci, v = synth.banded(rows, 60, 150)
    assert run(rp, ci, v, rows, rows) == (10, 0, 0)                       # every panel through LDS
    ci2, v2 = ci.copy(), v.copy()
    ci2[rp[100]:rp[101]] = ci2[rp[100]:rp[101]][::-1]                     # one descending row in panel 2
    v2[rp[100]:rp[101]] = v2[rp[100]:rp[101]][::-1]
    assert run(rp, ci2, v2, rows, rows) == (9, 0, 1)                      # that panel recomputed, others not
    rp3, ci3, v3 = synth.random_csr(rows, 5000, 30, seed=3, sorted_rows=True)
    assert run(rp3, ci3, v3, rows, 5000) == (0, 10, 0)                    # too sparse over its span: direct kernel
    rp3, ci3, v3 = synth.random_csr(rows, 5000, 6, seed=3, sorted_rows=True)
    assert run(rp3, ci3, v3, rows, 5000) == (0, 0, 0)                     # short rows throughout at 64 columns: not even classified
    rp4, ci4, v4 = synth.banded(rows, 500, 400)                           # rows of 500: several windows per (row, tile) visit
    assert run(rp4, ci4, v4, rows, rows) == (10, 0, 0)


# ---------------------------------------------------------------------------------------------------------
# skewed row lengths (webbase-like: the reference authors' own SpMV profiling input, profiling.sh:16,21)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["auto", "plain", "stream", "seg4", "lds"])
def test_spmv_power_law_rows(env, spmv_variant_env, variant):
    """Most rows hold 1-3 entries, a few hold thousands: the kernels are picked by the AVERAGE row length, so each
    must set its long rows aside (the lanes-per-row kernel: the whole block sums them; the stream kernel: a wave)."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    spmv_variant_env(variant)
    M = 30011
    rp, ci, v = synth.powerlaw(M, max_len=4000)
    assert np.diff(rp).max() == 4000 and rp[-1] / M < 5
    A = Dev(torch, dev, rp, ci, v, M)
    rng = np.random.default_rng(6)
    xh, yh = rng.standard_normal(M), rng.standard_normal(M)
    for alpha, beta in ((1.0, 1.0), (2.5, 0.0)):
        x, y = torch.from_numpy(xh).to(dev), torch.from_numpy(yh.copy()).to(dev)
        sblas.spmv(M, M, A.rowptr, A.colidx, A.val, x, alpha, beta, y)
        assert close(y.cpu().numpy(), oracle.spmv(M, *A.h, xh, yh.copy(), alpha, beta)), (variant, alpha, beta)


@pytest.mark.parametrize("variant", ["auto", "rows", "dpp"])
@pytest.mark.parametrize("n", [8, 33, 64, 130])
def test_spmm_power_law_rows(env, variant_env, variant, n):
    """The same rows through the SpMM direct kernels: the four-rows-per-wave kernel hands rows of 512+ entries to the
    whole workgroup (sixteen slices, merged in LDS); several long rows in one 64-row panel, a long row as the last
    row of the matrix, and a panel that is all long rows."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    variant_env(variant)
    M = 5003
    rp, ci, v = synth.powerlaw(M, max_len=3000)
    lens = np.diff(rp).astype(np.int64)
    lens[[70, 71, 100, M - 1]] = (600, 3000, 513, 1500)            # two long rows side by side, one just over the limit, the last row
    lens[128:192] = 520                                            # a whole panel of long rows
    rp2 = np.zeros(M + 1, np.int64)
    np.cumsum(lens, out=rp2[1:])
    rng = np.random.default_rng(8)
    ci2 = np.empty(rp2[-1], np.int32)
    for r in range(M):
        ci2[rp2[r]:rp2[r + 1]] = np.sort(rng.choice(M, lens[r], replace=False))
    v2 = rng.standard_normal(rp2[-1])
    A = Dev(torch, dev, rp2.astype(np.int32), ci2, v2, M)
    B, C0 = rng.standard_normal(M * n), rng.standard_normal(M * n)
    got = gpu_spmm(sblas, torch, dev, A, B, M, n, 1.5, -0.5, C0, M)
    ref = oracle.spmm(M, M, n, *A.h, B, C0.copy(), 1.5, -0.5)
    assert close(got, ref), (variant, n, np.abs(got - ref).max())


@pytest.mark.parametrize("variant", ["auto", "dpp"])
@pytest.mark.parametrize("n", [16, 64, 128, 256])
def test_spmm_row_per_wave_kernel_splits_very_long_rows(env, variant_env, variant, n):
    """The row-per-wave direct kernel (every width: 32- / 64- / 128-column tiles) hands rows of 4096+ entries to the
    whole workgroup: sixteen slices, partial sums added in LDS.  Rows average 40 entries (so that `auto` picks this
    kernel, not the four-rows-per-wave one) with a tail: one row just under the limit, one just over, one of 20 000
    entries with duplicates, two long rows in one 16-row panel, a long last row."""
    sblas, oracle, torch, dev = env
    variant_env(variant)
    M, K = 1203, 30000
    rng = np.random.default_rng(n)
    lens = rng.integers(30, 50, M).astype(np.int64)
    lens[[5, 17, 40, 320, 321, M - 1]] = (4095, 4096, 20000, 5000, 4500, 7001)
    lens[100] = 0
    rp = np.zeros(M + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    ci = np.empty(rp[-1], np.int32)
    for r in range(M):
        ci[rp[r]:rp[r + 1]] = np.sort(rng.integers(0, K, lens[r]))      # (with replacement: duplicates in the long rows)
    v = rng.standard_normal(rp[-1])
    A = Dev(torch, dev, rp.astype(np.int32), ci, v, K)
    B, C0 = rng.standard_normal(K * n), rng.standard_normal(M * n)
    sblas.panel_census()
    got = gpu_spmm(sblas, torch, dev, A, B, K, n, 1.5, -0.5, C0, M)
    census = sblas.panel_census()
    ref = oracle.spmm(M, K, n, *A.h, B, C0.copy(), 1.5, -0.5)
    assert close(got, ref), (variant, n, np.abs(got - ref).max())
    if variant == "auto":
        assert census["direct"] > 0, census


# ---------------------------------------------------------------------------------------------------------
# the matrix-core (MFMA) kernel: panels whose nonzeros sit in dense 16 x 4 sub-blocks
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 20, 64, 100, 128, 200])
@pytest.mark.parametrize("variant", ["auto", "mfma", "nomfma"])
def test_spmm_mfma_block_structured(env, variant_env, variant, n):
    """nd24k-like rows in dense sub-blocks (85 % fill; the bar against the LDS-tiled kernel is 68 %): the classifier samples the block fill and sends the panels to
    the fp64 MFMA kernel when the call is at least 128 staged columns wide; `mfma` forces it for every panel and
    width, `nomfma` forbids it.  Within 1e-10 relative of the oracle
    (observed: the MFMA adds a row's products in column order like the oracle's loop, zeros in between: <= 4e-16)."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    variant_env(variant)
    rows = 1000
    rp, ci, v = synth.block_structured(rows, nnz_per_row=150, half_band=400, fill=0.85)
    A = Dev(torch, dev, rp, ci, v, rows)
    rng = np.random.default_rng(n)
    B, C0 = rng.standard_normal(rows * n), rng.standard_normal(rows * n)
    sblas.panel_census()
    for alpha, beta in ((1.0, 1.0), (-0.5, 0.0)):
        got = gpu_spmm(sblas, torch, dev, A, B, rows, n, alpha, beta, C0, rows)
        ref = oracle.spmm(rows, rows, n, *A.h, B, C0.copy(), alpha, beta)
        assert close(got, ref), (variant, n, np.abs(got - ref).max())
    census = sblas.panel_census()
    if n > 32:   # (narrower blocks run on 8 / 16 / 32 staged columns: no matrix-core kernel there)
        # auto: the matrix cores from 128 staged columns on (a chunk step's fixed cost needs 8+ MFMAs per block to pay)
        assert (census["mfma"] > 0) == (variant == "mfma" or (variant == "auto" and n > 64)), census
        if variant == "mfma":
            assert census["windowed"] == 0 and census["direct"] == 0, census


@pytest.mark.parametrize("kind", ["random_unsorted", "duplicates", "ragged_rect", "one_long_row", "wide_span"])
def test_spmm_mfma_forced_on_any_matrix(env, variant_env, kind):
    """The MFMA kernel assumes nothing about the matrix (it merges the rows of a 16-row group chunk by chunk; LDS
    atomics add duplicates): forced onto unsorted rows, duplicate columns, empty rows, a 5000-entry row, a rectangular
    matrix whose rows are not a multiple of 16 and rows that span 100 000 columns."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    variant_env("mfma")
    n = 64
    if kind == "random_unsorted":
        rows, cols = 333, 777
        rp, ci, v = synth.random_csr(rows, cols, 20, seed=3, empty_every=7)
    elif kind == "duplicates":
        rows, cols = 200, 64
        rp, ci, v = synth.random_csr(rows, cols, 90, seed=4, sorted_rows=True)      # 90 draws from 64 columns
    elif kind == "ragged_rect":
        rows, cols = 1003, 5
        rp, ci, v = synth.random_csr(rows, cols, 3, seed=5, sorted_rows=True, empty_every=4)
    elif kind == "one_long_row":
        rows, cols = 100, 3000
        rp, ci, v = synth.random_csr(rows, cols, 4, seed=6, sorted_rows=True, long_row=(50, 5000))
    else:
        rows = cols = 150000
        rp, ci, v = synth.queen_like_grid(rows)
        rows = cols = len(rp) - 1
    A = Dev(torch, dev, rp, ci, v, cols)
    rng = np.random.default_rng(1)
    B, C0 = rng.standard_normal(cols * n), rng.standard_normal(rows * n)
    sblas.panel_census()
    got = gpu_spmm(sblas, torch, dev, A, B, cols, n, 2.0, -1.0, C0, rows)
    census = sblas.panel_census()
    if kind == "wide_span":
        ref = C0.copy()
        for r0 in (0, rows // 2, rows - 64):
            oracle.spmm_rows(r0, r0 + 64, rows, cols, n, *A.h, B, ref, 2.0, -1.0)
            assert close(got.reshape(n, rows)[:, r0:r0 + 64], ref.reshape(n, rows)[:, r0:r0 + 64]), r0
    else:
        ref = oracle.spmm(rows, cols, n, *A.h, B, C0.copy(), 2.0, -1.0)
        assert close(got, ref), (kind, np.abs(got - ref).max())
    if kind in ("duplicates", "ragged_rect", "one_long_row", "wide_span"):      # sorted rows: every panel took the matrix cores
        assert census["mfma"] > 0 and census["direct"] == 0 and census["windowed"] == 0, census


def test_spmm_mfma_leaves_nonfinite_b_to_the_vector_kernels(env, variant_env):
    """0 * Inf from a block's zero fill would reach rows that never refer to that row of B, so a staging pass that met
    a non-finite value hands the MFMA panels to the vector kernels: rows that do not refer to the Inf row stay finite
    and exact, rows that do carry the oracle's Inf / NaN."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    variant_env("mfma")
    rows, n = 640, 64
    rp, ci, v = synth.block_structured(rows, nnz_per_row=100, half_band=300, fill=0.7)
    A = Dev(torch, dev, rp, ci, v, rows)
    rng = np.random.default_rng(2)
    Bh, C0 = rng.standard_normal(rows * n), rng.standard_normal(rows * n)
    Bh.reshape(n, rows)[3, int(ci[rp[100]])] = np.inf            # one Inf in a referenced row of B
    sblas.panel_census()
    got = gpu_spmm(sblas, torch, dev, A, Bh, rows, n, 1.0, 1.0, C0, rows)
    census = sblas.panel_census()
    ref = oracle.spmm(rows, rows, n, *A.h, Bh, C0.copy(), 1.0, 1.0)
    fin = np.isfinite(ref)
    assert (np.isfinite(got) == fin).all()
    assert close(got[fin], ref[fin])
    assert census["mfma"] == 0 and census["windowed"] + census["direct"] > 0, census
    # the same workspace with a finite B afterwards: back on the matrix cores
    Bh2 = rng.standard_normal(rows * n)
    got2 = gpu_spmm(sblas, torch, dev, A, Bh2, rows, n, 1.0, 1.0, C0, rows)
    assert close(got2, oracle.spmm(rows, rows, n, *A.h, Bh2, C0.copy(), 1.0, 1.0))
    assert sblas.panel_census()["mfma"] > 0


@pytest.fixture
def spmv_variant_env():
    yield from _env_switch("SBLAS_SPMV_VARIANT")


@pytest.mark.parametrize("variant", ["plain", "lds", "lds2", "lds1s2", "lds1s3", "lds1s4", "auto", "seg2", "seg3", "seg4", "seg8", "stream"])
@pytest.mark.parametrize("kind", ["banded", "unsorted", "wide_span", "outliers"])
def test_spmv_long_rows_any_structure(env, spmv_variant_env, variant, kind):
    """Long rows (the 64-lanes-per-row instantiation, unrolled four slices deep): banded, shuffled, very wide spans
    and far-away outlier columns; the default kernel and the two experimental long-row kernels (one stages the x
    window in LDS and must fall back lane by lane for columns outside it)."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    spmv_variant_env(variant)
    rng = np.random.default_rng(5)
    if kind == "wide_span":
        M = K = 60000
        rp, ci, v = synth.banded(M, 100, 20000)          # span 40 000 > LDS window
    else:
        M = K = 5000
        rp, ci, v = synth.banded(M, 120, 700)
    ci, v = ci.copy(), v.copy()
    if kind == "unsorted":
        for r in range(0, M, 3):
            perm = rng.permutation(rp[r + 1] - rp[r])
            ci[rp[r]:rp[r + 1]] = ci[rp[r]:rp[r + 1]][perm]
            v[rp[r]:rp[r + 1]] = v[rp[r]:rp[r + 1]][perm]
    if kind == "outliers":
        ci[rp[:-1][::5] + 7] = rng.integers(0, K, len(rp[:-1][::5]))       # one far-away column in every 5th row
    A = Dev(torch, dev, rp, ci, v, K)
    xh, yh = rng.standard_normal(K), rng.standard_normal(M)
    for alpha, beta in ((1.0, 1.0), (-0.5, 0.0)):
        x, y = torch.from_numpy(xh).to(dev), torch.from_numpy(yh.copy()).to(dev)
        sblas.spmv(M, K, A.rowptr, A.colidx, A.val, x, alpha, beta, y)
        ref = oracle.spmv(M, *A.h, xh, yh.copy(), alpha, beta)
        assert close(y.cpu().numpy(), ref), (kind, alpha, beta)


@pytest.mark.parametrize("variant", ["auto", "lds", "lds1s2", "lds1s3", "lds2"])
def test_spmv_unstaged_window_never_multiplies_stale_lds(env, spmv_variant_env, variant):
    """a 16-row block whose rows span more columns than the LDS window stages nothing; its clamped lanes
    must not multiply by whatever an earlier kernel left in LDS.  The staging kernel run first leaves NaN tiles at
    the start of every CU's LDS; row lengths are not multiples of 64 so that every row has clamped lanes, and x
    holds an Inf at a column only the LAST entry of a row refers to (0 * Inf in a clamped lane would be a NaN)."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    spmv_variant_env(variant)
    M = K = 30000
    rp, ci, v = synth.banded(M, 101, 9000)            # span 18 000 columns > 5120: nothing staged
    rp2, ci2, v2 = synth.banded(4000, 133, 1500)      # span 3000: staged, last column of row 7 gets an Inf in x
    nanB = torch.full((4096 * 64,), float("nan"), dtype=torch.float64, device=dev)
    Bt = torch.empty(sblas.spmm_workspace_bytes(1, 4096, 1, 64) // 8, dtype=torch.float64, device=dev)
    for (m, k, rp_, ci_, v_) in ((M, K, rp, ci, v), (4000, 4000, rp2, ci2, v2)):
        A = Dev(torch, dev, rp_, ci_, v_, k)
        xh = np.random.default_rng(8).standard_normal(k)
        if m == 4000:
            last = int(ci_[rp_[8] - 1])
            only_last = not np.isin(last, np.delete(ci_, rp_[8] - 1))
            if only_last:
                xh[last] = np.inf
        yh = np.random.default_rng(9).standard_normal(m)
        sblas.dense_to_rowmajor(4096, 64, nanB, 4096, Bt)                     # NaN tiles in every CU's LDS
        x, y = torch.from_numpy(xh).to(dev), torch.from_numpy(yh.copy()).to(dev)
        sblas.spmv(m, k, A.rowptr, A.colidx, A.val, x, 1.5, -0.5, y)
        got = y.cpu().numpy()
        ref = oracle.spmv(m, *A.h, xh, yh.copy(), 1.5, -0.5)
        fin = np.isfinite(ref)
        assert (np.isfinite(got) == fin).all() and not np.isnan(got).any(), variant
        assert close(got[fin], ref[fin]) and (got[~fin] == ref[~fin]).all()


@pytest.mark.parametrize("xoff", [0, 1])
@pytest.mark.parametrize("K", [4000, 4001])
def test_spmv_lds_window_fetch_paths(env, xoff, K):
    """The long-row kernel fetches the x window by LDS-DMA in 16-byte pieces when x is 16-byte aligned and the last
    pair lies inside x, otherwise through registers: an x that starts 8 bytes off, an odd column count with rows
    that reach the last column, and the aligned case."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    M = 3000
    rp, ci, v = synth.banded(M, 200, 900)
    ci = np.minimum(ci.astype(np.int64) + (K - M), K - 1).astype(np.int32)     # push the band against the last column
    for r in range(M):                                                         # (clipping made duplicates: fine, keep order)
        ci[rp[r]:rp[r + 1]].sort()
    A = Dev(torch, dev, rp, ci, v, K)
    rng = np.random.default_rng(K + xoff)
    xh, yh = rng.standard_normal(K), rng.standard_normal(M)
    xbuf = torch.zeros(K + 2, dtype=torch.float64, device=dev)
    x = xbuf[xoff:xoff + K]
    x.copy_(torch.from_numpy(xh))
    y = torch.from_numpy(yh.copy()).to(dev)
    sblas.spmv(M, K, A.rowptr, A.colidx, A.val, x, 1.5, -1.0, y)
    assert close(y.cpu().numpy(), oracle.spmv(M, *A.h, xh, yh.copy(), 1.5, -1.0))


@pytest.mark.parametrize("variant", ["seg2", "seg3", "seg4", "seg8", "stream"])
@pytest.mark.parametrize("avg", [2, 30, 73, 150])
def test_spmv_segmented_rows_per_wave(env, spmv_variant_env, variant, avg):
    """The segmented kernel (R rows per wave, one contiguous run of nonzeros, per-row accumulation by row pointer
    comparison): empty rows, a row far longer than a burst, unsorted columns, row counts that are not a multiple of R."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    spmv_variant_env(variant)
    M, K = 1237, 4000
    rp, ci, v = synth.random_csr(M, K, avg, seed=avg, empty_every=7, long_row=(33, 3000))
    A = Dev(torch, dev, rp, ci, v, K)
    rng = np.random.default_rng(avg)
    xh, yh = rng.standard_normal(K), rng.standard_normal(M)
    for alpha, beta in ((1.0, 1.0), (2.0, 0.0)):
        x, y = torch.from_numpy(xh).to(dev), torch.from_numpy(yh.copy()).to(dev)
        sblas.spmv(M, K, A.rowptr, A.colidx, A.val, x, alpha, beta, y)
        assert close(y.cpu().numpy(), oracle.spmv(M, *A.h, xh, yh.copy(), alpha, beta)), (variant, avg, alpha, beta)


@pytest.mark.parametrize("cap", [6144, 4096])
@pytest.mark.parametrize("variant", ["stream", "auto"])
def test_spmv_stream_runs_and_oversize_rows(env, spmv_variant_env, variant, cap):
    """The short-row stream kernel parks a block's products in LDS (6144 or 4096 per run; the launcher takes the capacity
    that gives a block of average rows the fewest runs, round 3): a block whose 256 rows hold more is taken in several
    runs, and a single row beyond the capacity is summed by the whole block.  cap 6144: rows of 21 on average with one
    block of 40-nonzero rows, one row of 7000 and one of 6144 exactly; cap 4096: rows of 11 on average, a block of
    30-nonzero rows, one row of 5000 and one of 4096 exactly.  The row count is not a multiple of 256."""
    sblas, oracle, torch, dev = env
    spmv_variant_env(variant)
    rng = np.random.default_rng(5)
    M, K = 2000, 9000
    if cap == 6144:
        lens = rng.integers(0, 20, M)
        lens[300:560] = 40
        lens[700] = 7000
        lens[1500] = 6144
    else:
        lens = rng.integers(0, 8, M)
        lens[300:560] = 30
        lens[700] = 5000
        lens[1500] = 4096
    assert (256.0 * lens.sum() / M <= 4096) == (cap == 4096)      # which instantiation the launcher takes
    lens[M - 1] = 3
    rp = np.zeros(M + 1, dtype=np.int32)
    rp[1:] = np.cumsum(lens)
    ci = np.concatenate([rng.choice(K, l, replace=False) for l in lens]).astype(np.int32)
    v = rng.standard_normal(len(ci))
    A = Dev(torch, dev, rp, ci, v, K)
    xh, yh = rng.standard_normal(K), rng.standard_normal(M)
    for alpha, beta in ((1.0, 1.0), (2.0, 0.0)):
        x, y = torch.from_numpy(xh).to(dev), torch.from_numpy(yh.copy()).to(dev)
        sblas.spmv(M, K, A.rowptr, A.colidx, A.val, x, alpha, beta, y)
        assert close(y.cpu().numpy(), oracle.spmv(M, *A.h, xh, yh.copy(), alpha, beta)), (variant, alpha, beta)


def test_spmm_kernel_event_hook(env):
    """sblas_hip_debug_spmm_kernel_events: the launcher brackets the dominant stage-2 kernel with HIP events while the
    hook is on (bench.py times its roofline object with it); off again, launches record nothing new."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows, n = 3000, 64
    rp, ci, v = synth.banded(rows, 120, 500)
    A = Dev(torch, dev, rp, ci, v, rows)
    B = torch.rand(rows * n, dtype=torch.float64, device=dev)
    C = torch.zeros(rows * n, dtype=torch.float64, device=dev)
    ws = torch.empty(sblas.spmm_workspace_bytes(rows, rows, len(ci), n) // 8, dtype=torch.float64, device=dev)
    sblas.kernel_events(True)
    try:
        sblas.spmm(rows, rows, A.rowptr, A.colidx, A.val, B, rows, n, 1.0, 0.0, C, rows, ws)
        ms = sblas.last_kernel_ms()
        assert 0.0 < ms < 50.0
    finally:
        sblas.kernel_events(False)
    ref = oracle.spmm(rows, rows, n, *A.h, B.cpu().numpy(), np.zeros(rows * n), 1.0, 0.0)
    assert close(C.cpu().numpy(), ref)


@pytest.mark.parametrize("variant", ["auto", "merge"])
@pytest.mark.parametrize("start", [0, 1, 2, 3, 1000, 1001, 1002])
def test_spmm_row_merging_kernel_on_row_blocks_at_any_offset(env, variant_env, variant, start):
    """Three rows per mesh node share one column pattern; a method-2 row block starts at any row, so the groups of
    three start at any offset modulo 3 inside the block.  The row-merging kernel lines its waves up with the groups
    (phase from the classifier); the result never depends on it.  Also a block whose LAST rows are a cut-off group."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    variant_env(variant)
    rp, ci, v = synth.queen_like_grid(9000, half_band=1500)
    K = len(rp) - 1
    rng = np.random.default_rng(start)
    for m, n in ((4001, 128), (2500, 256)):
        a, b = start, start + m
        sub = (rp[a:b + 1] - rp[a]).astype(np.int32)
        A = Dev(torch, dev, sub, ci[rp[a]:rp[b]], v[rp[a]:rp[b]], K)
        B, C0 = rng.standard_normal(K * n), rng.standard_normal(m * n)
        sblas.panel_census()
        got = gpu_spmm(sblas, torch, dev, A, B, K, n, 0.5, 2.0, C0, m)
        assert variant != "auto" or sblas.panel_census()["direct"] > 0      # (the forced kernel does not count panels)
        assert close(got, oracle.spmm(m, K, n, *A.h, B, C0.copy(), 0.5, 2.0))


@pytest.mark.parametrize("dofs", [2, 3, 6])
def test_spmm_row_merging_kernel_only_where_three_rows_share_a_pattern(env, dofs):
    """Nodes of 3 (and 6 = 2 x 3) unknowns give groups of three rows with one column pattern: the row-merging kernel
    owns the call's direct panels.  Nodes of 2 give pairs only: that kernel would run its slower unmerged path, so the
    call stays with the row-per-wave kernel.  Read back from the workspace header (TAIL_DIRECT_EPOCH = 2,
    TAIL_MERGE_EPOCH = 6 in kernels.h); the result is checked either way, for a whole matrix and for an odd row block."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rp, ci, v = synth.queen_like_grid(24000, half_band=9000, dofs=dofs)        # wide band: no panel fits the LDS-tiled kernel
    K = len(rp) - 1
    n = 128
    rng = np.random.default_rng(dofs)
    for a, b in ((0, K), (1001, 4500)):
        m = b - a
        sub = (rp[a:b + 1] - rp[a]).astype(np.int32)
        A = Dev(torch, dev, sub, ci[rp[a]:rp[b]], v[rp[a]:rp[b]], K)
        Bh, C0 = rng.standard_normal(K * n), rng.standard_normal(m * n)
        B, C = torch.from_numpy(Bh).to(dev), torch.from_numpy(C0.copy()).to(dev)
        ws = torch.zeros(sblas.spmm_workspace_bytes(m, K, len(A.h[1]), n) // 8, dtype=torch.float64, device=dev)
        sblas.panel_census()
        sblas.spmm(m, K, A.rowptr, A.colidx, A.val, B, K, n, 2.0, -1.0, C, m, ws)
        torch.cuda.synchronize()
        assert sblas.panel_census()["direct"] > 0
        hdr = ws[(K + 1) * 128:].view(torch.int32)[:16].cpu().numpy()
        assert hdr[2] != 0
        assert (hdr[6] == hdr[2]) == (dofs != 2), (dofs, a, hdr[:8])
        assert close(C.cpu().numpy(), oracle.spmm(m, K, n, *A.h, Bh, C0.copy(), 2.0, -1.0))


@pytest.mark.parametrize("n", [64, 200, 256, 300])
def test_spmm_column_chunking_when_bt_exceeds_the_offset_window(env, n):
    """BASELINE config 5 shape problem (Queen_4147, N = 256: the row-major copy of B is 8.5 GB) in miniature: with the
    Bt byte limit lowered, the top-level call must walk the dense columns in chunks (128 / 64 / 32 wide) and still
    match the oracle; the workspace query must shrink accordingly."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    M = K = 700
    rp, ci, v = synth.banded(M, 30, 60)
    A = Dev(torch, dev, rp, ci, v, K)
    rng = np.random.default_rng(n)
    B, C0 = rng.standard_normal(K * n), rng.standard_normal(M * n)
    ref = oracle.spmm(M, K, n, *A.h, B, C0.copy(), 1.5, -1.0)
    full = sblas.spmm_workspace_bytes(M, K, len(ci), n)
    for limit, width in ((8 * 701 * 128, 128), (8 * 701 * 64, 64), (8 * 701 * 40, 32)):
        os.environ["SBLAS_SPMM_MAX_BT_BYTES"] = str(limit)
        sblas.reload_env()
        try:
            ws = sblas.spmm_workspace_bytes(M, K, len(ci), n)
            if n > width:
                assert ws < full
            got = gpu_spmm(sblas, torch, dev, A, B, K, n, 1.5, -1.0, C0, M)
        finally:
            os.environ.pop("SBLAS_SPMM_MAX_BT_BYTES", None)
            sblas.reload_env()
        assert close(got, ref), (n, width, np.abs(got - ref).max())
    # the same walk with the range staging of a row block: every chunk stages the block's column range again
    sub = (rp[200:451] - rp[200]).astype(np.int32)
    As = Dev(torch, dev, sub, ci[rp[200]:rp[450]], v[rp[200]:rp[450]], K)
    Cs = rng.standard_normal(250 * n)
    want = oracle.spmm(250, K, n, *As.h, B, Cs.copy(), 1.5, -1.0)
    os.environ["SBLAS_SPMM_MAX_BT_BYTES"] = str(8 * 701 * 64)
    os.environ["SBLAS_STAGE_RANGE"] = "1"
    sblas.reload_env()
    try:
        got = gpu_spmm(sblas, torch, dev, As, B, K, n, 1.5, -1.0, Cs, 250)
    finally:
        os.environ.pop("SBLAS_SPMM_MAX_BT_BYTES", None)
        os.environ.pop("SBLAS_STAGE_RANGE", None)
        sblas.reload_env()
    assert close(got, want)


@pytest.mark.parametrize("variant", ["auto", "dpp"])
def test_spmm_queen_like_n256(env, variant_env, variant):
    """BASELINE config 5 shape (Queen_4147-like stencil rows, N = 256) at reduced size: wide spans send every panel to
    the direct kernel, 128-column tiles, four column tiles."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    variant_env(variant)
    rows, n = 6000, 256
    rp, ci, v = synth.queen_like(rows, half_band=2500)
    assert 55 < len(ci) / rows < 95 and all((np.diff(ci[rp[r]:rp[r + 1]]) > 0).all() for r in (0, 1, 2999, rows - 1))
    A = Dev(torch, dev, rp, ci, v, rows)
    rng = np.random.default_rng(8)
    B, C0 = rng.standard_normal(rows * n), rng.standard_normal(rows * n)
    sblas.panel_stats()
    got = gpu_spmm(sblas, torch, dev, A, B, rows, n, 1.0, 1.0, C0, rows)
    st = sblas.panel_stats()
    ref = oracle.spmm(rows, rows, n, *A.h, B, C0.copy(), 1.0, 1.0)
    assert close(got, ref)
    if variant == "auto":
        assert st[0] == 0 and st[1] > 0 and st[2] == 0        # nothing windowed, nothing fell back


@pytest.mark.parametrize("kind", ["banded", "mixed"])
def test_spmm_and_spmv_inside_a_hip_graph(env, kind):
    """The product path is stream-ordered only (no host synchronisation, no allocation, no host read-back): a call
    can be captured into a HIP graph and replayed on new values in the same buffers.  Banded rows take the LDS-tiled
    kernel, the mixed matrix sends most panels to the direct kernel."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows, n = 4000, 64
    if kind == "banded":
        rp, ci, v = synth.banded(rows, 150, 600)
    else:
        rp, ci, v = synth.random_csr(rows, rows, 40, seed=3, sorted_rows=True, empty_every=11, long_row=(17, 2500))
    A = Dev(torch, dev, rp, ci, v, rows)
    rng = np.random.default_rng(11)
    B = torch.from_numpy(rng.standard_normal(rows * n)).to(dev)
    C = torch.zeros(rows * n, dtype=torch.float64, device=dev)
    x = torch.from_numpy(rng.standard_normal(rows)).to(dev)
    y = torch.zeros(rows, dtype=torch.float64, device=dev)
    ws = torch.empty(sblas.spmm_workspace_bytes(rows, rows, len(ci), n) // 8, dtype=torch.float64, device=dev)
    sblas.spmm(rows, rows, A.rowptr, A.colidx, A.val, B, rows, n, 1.0, 0.0, C, rows, ws)   # warm-up outside the capture
    sblas.spmv(rows, rows, A.rowptr, A.colidx, A.val, x, 1.0, 0.0, y)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sblas.spmm(rows, rows, A.rowptr, A.colidx, A.val, B, rows, n, 2.0, 0.0, C, rows, ws)
        sblas.spmv(rows, rows, A.rowptr, A.colidx, A.val, x, 2.0, 0.0, y)
    for rep in range(2):                                       # new values in the captured buffers, then replay
        Bh, xh = rng.standard_normal(rows * n), rng.standard_normal(rows)
        B.copy_(torch.from_numpy(Bh).to(dev))
        x.copy_(torch.from_numpy(xh).to(dev))
        C.fill_(7.0)
        y.fill_(7.0)
        g.replay()
        torch.cuda.synchronize()
        assert close(C.cpu().numpy(), oracle.spmm(rows, rows, n, *A.h, Bh, np.zeros(rows * n), 2.0, 0.0)), (kind, rep)
        assert close(y.cpu().numpy(), oracle.spmv(rows, *A.h, xh, np.zeros(rows), 2.0, 0.0)), (kind, rep)


@pytest.mark.parametrize("n", [1, 3, 8])
@pytest.mark.parametrize("kind", ["banded", "ragged"])
def test_spmm_up_to_eight_columns_long_rows(env, n, kind):
    """N <= 8 with long rows takes the wave-per-row kernel (eight sums per lane, halving exchange) from 256 nonzeros
    per row on average; the test moves the switch-over down with SBLAS_ROWS8_MIN_AVG.  Row counts that are not a multiple of 4, empty rows, one row of 3000, lengths that are not a multiple of
    64 or 256; an Inf in a row of B that no nonzero refers to must not leak in through the padding lanes."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    for setter in _env_switch("SBLAS_ROWS8_MIN_AVG"):
        setter("96")
        _eight_columns_long_rows(sblas, oracle, torch, dev, synth, n, kind)


def _eight_columns_long_rows(sblas, oracle, torch, dev, synth, n, kind):
    if kind == "banded":
        rows = 2999
        rp, ci, v = synth.banded(rows, 150, 700)
        cols = rows
    else:
        rows, cols = 1237, 4000
        rp, ci, v = synth.random_csr(rows, cols, 130, seed=7, empty_every=9, long_row=(33, 3000))
    A = Dev(torch, dev, rp, ci, v, cols)
    rng = np.random.default_rng(n)
    Bh = rng.standard_normal(cols * n)
    unused = np.setdiff1d(np.arange(cols), ci)
    if unused.size:
        Bh.reshape(n, cols)[:, unused[0]] = np.inf
    Ch = rng.standard_normal(rows * n)
    ws = torch.empty(max(sblas.spmm_workspace_bytes(rows, cols, len(ci), n) // 8, 1), dtype=torch.float64, device=dev)
    for alpha, beta in ((1.0, 1.0), (-2.0, 0.0), (0.5, 3.0)):
        B, C = torch.from_numpy(Bh).to(dev), torch.from_numpy(Ch.copy()).to(dev)
        sblas.spmm(rows, cols, A.rowptr, A.colidx, A.val, B, cols, n, alpha, beta, C, rows, ws)
        ref = oracle.spmm(rows, cols, n, *A.h, Bh, Ch.copy(), alpha, beta)
        assert close(C.cpu().numpy(), ref), (n, kind, alpha, beta)


def test_spmm_and_spmv_from_two_streams_at_once(env):
    """Two streams of one device issue calls side by side (own workspace each, no synchronisation in between): the
    library keeps no per-call state outside the workspace and the epoch counter is atomic, so the calls must not
    disturb each other -- an LDS-tiled matrix on one stream, a direct-path matrix and an SpMV on the other, ten rounds
    with fresh B."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    n = 64
    rp1, ci1, v1 = synth.banded(3000, 120, 400)
    rp2, ci2, v2 = synth.random_csr(2500, 2500, 30, seed=5, sorted_rows=True, long_row=(9, 2000))
    A1, A2 = Dev(torch, dev, rp1, ci1, v1, 3000), Dev(torch, dev, rp2, ci2, v2, 2500)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ws1 = torch.empty(sblas.spmm_workspace_bytes(3000, 3000, len(ci1), n) // 8, dtype=torch.float64, device=dev)
    ws2 = torch.empty(sblas.spmm_workspace_bytes(2500, 2500, len(ci2), n) // 8, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(12)
    B1h = [rng.standard_normal(3000 * n) for _ in range(10)]
    B2h = [rng.standard_normal(2500 * n) for _ in range(10)]
    B1 = [torch.from_numpy(b).to(dev) for b in B1h]
    B2 = [torch.from_numpy(b).to(dev) for b in B2h]
    C1 = [torch.zeros(3000 * n, dtype=torch.float64, device=dev) for _ in range(10)]
    C2 = [torch.zeros(2500 * n, dtype=torch.float64, device=dev) for _ in range(10)]
    y2 = [torch.zeros(2500, dtype=torch.float64, device=dev) for _ in range(10)]
    torch.cuda.synchronize()
    for k in range(10):
        sblas.spmm(3000, 3000, A1.rowptr, A1.colidx, A1.val, B1[k], 3000, n, 1.0, 0.0, C1[k], 3000, ws1, stream=s1)
        sblas.spmm(2500, 2500, A2.rowptr, A2.colidx, A2.val, B2[k], 2500, n, 2.0, 0.0, C2[k], 2500, ws2, stream=s2)
        sblas.spmv(2500, 2500, A2.rowptr, A2.colidx, A2.val, B2[k][:2500], 1.0, 0.0, y2[k], stream=s2)
    torch.cuda.synchronize()
    for k in (0, 4, 9):
        assert close(C1[k].cpu().numpy(), oracle.spmm(3000, 3000, n, *A1.h, B1h[k], np.zeros(3000 * n), 1.0, 0.0)), k
        assert close(C2[k].cpu().numpy(), oracle.spmm(2500, 2500, n, *A2.h, B2h[k], np.zeros(2500 * n), 2.0, 0.0)), k
        assert close(y2[k].cpu().numpy(), oracle.spmv(2500, *A2.h, B2h[k][:2500].copy(), np.zeros(2500), 1.0, 0.0)), k


def test_spmm_from_two_host_threads(env):
    """The reference drives every GPU from its own OpenMP thread (spmm.h:100-104); here two host threads call into the
    library at once (ctypes releases the GIL), each on its own stream and workspace: the process-wide state (options,
    epoch counter, per-kernel LDS limits) is guarded, results match the oracle."""
    import threading
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    n = 128
    mats = [synth.banded(4000, 150, 500, seed=1), synth.queen_like_grid(6000, half_band=2500)]
    jobs = []
    for rp, ci, v in mats:
        rows = len(rp) - 1
        A = Dev(torch, dev, rp, ci, v, rows)
        Bh = np.random.default_rng(rows).standard_normal(rows * n)
        jobs.append(dict(A=A, rows=rows, Bh=Bh, B=torch.from_numpy(Bh).to(dev), C=torch.zeros(rows * n, dtype=torch.float64, device=dev),
                         ws=torch.empty(sblas.spmm_workspace_bytes(rows, rows, len(ci), n) // 8, dtype=torch.float64, device=dev),
                         stream=torch.cuda.Stream(device=dev), err=None))
    torch.cuda.synchronize()

    def work(j):
        try:
            torch.cuda.set_device(dev)
            for _ in range(20):
                sblas.spmm(j["rows"], j["rows"], j["A"].rowptr, j["A"].colidx, j["A"].val, j["B"], j["rows"], n, 1.0, 0.0, j["C"],
                           j["rows"], j["ws"], stream=j["stream"])
        except Exception as ex:      # surfaced by the main thread
            j["err"] = ex

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    for j in jobs:
        assert j["err"] is None, j["err"]
        assert close(j["C"].cpu().numpy(), oracle.spmm(j["rows"], j["rows"], n, *j["A"].h, j["Bh"], np.zeros(j["rows"] * n), 1.0, 0.0))
