"""The per-matrix plan (sblas_hip_spmm_plan_*, the slot of cuSPARSE's bufferSize / workspace step at spmm.h:134-141): a
planned call must give bit-identical results to the unplanned call on every kernel selection, launch only the kernels
that have panels, follow new values in A and B, refuse a structure it was not made for, and stay graph-capturable."""
import numpy as np
import pytest

from test_gpu_parity import Dev, close, _env_switch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(sblas, oracle, cuda):
    import torch
    return sblas, oracle, torch, cuda


@pytest.fixture
def stage_range_env():
    yield from _env_switch("SBLAS_STAGE_RANGE")


def _shapes(synth):
    rng = np.random.default_rng(5)
    mixed_lens = np.where(np.arange(600) % 192 < 96, 150, 3).astype(np.int64)
    mixed_lens[10] = 0
    rp = np.zeros(601, np.int64)
    rp[1:] = np.cumsum(mixed_lens)
    ci = np.empty(rp[-1], np.int32)
    for r in range(600):
        if mixed_lens[r] >= 100:
            lo = min(r * 4, 3000 - 800)
            ci[rp[r]:rp[r + 1]] = np.sort(rng.choice(np.arange(lo, lo + 800), mixed_lens[r], replace=False))
        else:
            ci[rp[r]:rp[r + 1]] = np.sort(rng.choice(3000, mixed_lens[r], replace=False))
    mixed = (rp.astype(np.int32), ci, rng.standard_normal(rp[-1]))
    return {
        "banded": (synth.banded(1500, 80, 300), 1500),                       # every panel through LDS
        "sparse": (synth.random_csr(1000, 4000, 7, seed=3, sorted_rows=True, empty_every=9), 4000),   # every panel direct
        "mixed": (mixed, 3000),                                              # both
        "grid": (synth.queen_like_grid(3000, half_band=400), 3000),          # rows in groups of three: row merging at 128+
        "blocks": (synth.block_structured(1000, nnz_per_row=150, half_band=400, fill=0.85), 1000),     # matrix cores at 128+
    }


@pytest.mark.parametrize("n", [8, 16, 32, 64, 128, 200])
@pytest.mark.parametrize("shape", ["banded", "sparse", "mixed", "grid", "blocks"])
def test_planned_call_is_bit_identical_and_launches_only_what_has_panels(env, shape, n):
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    (rp, ci, v), cols = _shapes(synth)[shape]
    rows = len(rp) - 1
    A = Dev(torch, dev, rp, ci, v, cols)
    rng = np.random.default_rng(n)
    Bh, C0 = rng.standard_normal(cols * n), rng.standard_normal(rows * n)
    B = torch.from_numpy(Bh).to(dev)
    ws = torch.empty(sblas.spmm_workspace_bytes(rows, cols, len(ci), n) // 8, dtype=torch.float64, device=dev)
    plan = sblas.SpmmPlan(rows, cols, A.rowptr, A.colidx, n)
    info = plan.info()
    short_rows_narrow = shape == "sparse" and n <= 64              # such a call classifies nothing: nothing to plan either
    assert info["active"] == (not short_rows_narrow) and (short_rows_narrow or info["ldbt"] == int(sblas.lib().sblas_hip_spmm_ldbt(n)))
    for alpha, beta in ((1.5, -0.5), (1.0, 0.0)):
        Cu = torch.from_numpy(C0.copy()).to(dev)
        Cp = torch.from_numpy(C0.copy()).to(dev)
        sblas.panel_census()
        sblas.spmm(rows, cols, A.rowptr, A.colidx, A.val, B, cols, n, alpha, beta, Cu, rows, ws)
        cu = sblas.panel_census()
        plan.spmm(A.val, B, cols, n, alpha, beta, Cp, rows, ws)
        cp = sblas.panel_census()
        torch.cuda.synchronize()
        assert torch.equal(Cu, Cp), (shape, n, alpha, beta)               # same kernels on the same panels: same bits
        assert cu == cp, (cu, cp)
        assert close(Cp.cpu().numpy(), oracle.spmm(rows, cols, n, *A.h, Bh, C0.copy(), alpha, beta))
    # the plan's census is what the kernels then count
    nchunks = 1
    assert (info["windowed"] > 0) == (cp["windowed"] + cp["fallback"] > 0), (info, cp)
    assert (info["mfma"] > 0) == (cp["mfma"] > 0), (info, cp)
    if shape == "banded":
        assert info["direct"] == 0 and info["windowed"] > 0
    if shape == "sparse" and not short_rows_narrow:
        assert info["windowed"] == 0 and info["direct"] > 0
    if shape == "grid" and n >= 128:
        assert info["merge"]
    if shape == "blocks":
        assert (info["mfma"] > 0) == (n > 64)
    plan.destroy()


def test_plan_follows_new_values_and_refuses_another_structure(env):
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows, n = 1200, 64
    rp, ci, v = synth.banded(rows, 80, 300)
    A = Dev(torch, dev, rp, ci, v, rows)
    plan = sblas.SpmmPlan(rows, rows, A.rowptr, A.colidx, n)
    ws = torch.empty(sblas.spmm_workspace_bytes(rows, rows, len(ci), n) // 8, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(1)
    for rep in range(2):                                               # new values of A and B under the same plan
        vh, Bh = rng.standard_normal(len(ci)), rng.standard_normal(rows * n)
        val, B = torch.from_numpy(vh).to(dev), torch.from_numpy(Bh).to(dev)
        C = torch.zeros(rows * n, dtype=torch.float64, device=dev)
        plan.spmm(val, B, rows, n, 1.0, 0.0, C, rows, ws)
        torch.cuda.synchronize()
        assert close(C.cpu().numpy(), oracle.spmm(rows, rows, n, rp, ci, vh, Bh, np.zeros(rows * n), 1.0, 0.0))
    # a non-finite value in B: the flag lives in the plan's buffer for the call that staged it, and clears with the next
    Bh = rng.standard_normal(rows * n)
    Bbad = Bh.copy()
    Bbad[5] = np.inf
    C = torch.zeros(rows * n, dtype=torch.float64, device=dev)
    plan.spmm(A.val, torch.from_numpy(Bbad).to(dev), rows, n, 1.0, 0.0, C, rows, ws)
    plan.spmm(A.val, torch.from_numpy(Bh).to(dev), rows, n, 1.0, 0.0, C, rows, ws)
    torch.cuda.synchronize()
    assert close(C.cpu().numpy(), oracle.spmm(rows, rows, n, *A.h, Bh, np.zeros(rows * n), 1.0, 0.0))
    other = Dev(torch, dev, rp, ci, v, rows)                            # equal contents, other arrays: not this plan's
    with pytest.raises(sblas.SblasError):
        lib = sblas.lib()
        import ctypes as C_
        rc = lib.sblas_hip_spmm_csr_f64_i32_planned(plan.handle, -1, None, rows, rows, len(ci), C_.c_void_p(other.rowptr.data_ptr()),
                                                    C_.c_void_p(other.colidx.data_ptr()), C_.c_void_p(other.val.data_ptr()),
                                                    C_.c_void_p(C.data_ptr()), rows, n, 1.0, 0.0, C_.c_void_p(C.data_ptr()), rows,
                                                    C_.c_void_p(ws.data_ptr()), ws.numel() * 8)
        sblas.check(rc, "planned call on another structure")
    with pytest.raises(sblas.SblasError):                               # ... nor another device's
        rc = lib.sblas_hip_spmm_csr_f64_i32_planned(plan.handle, 7, None, rows, rows, len(ci), C_.c_void_p(A.rowptr.data_ptr()),
                                                    C_.c_void_p(A.colidx.data_ptr()), C_.c_void_p(A.val.data_ptr()),
                                                    C_.c_void_p(C.data_ptr()), rows, n, 1.0, 0.0, C_.c_void_p(C.data_ptr()), rows,
                                                    C_.c_void_p(ws.data_ptr()), ws.numel() * 8)
        sblas.check(rc, "planned call on another device")
    plan.destroy()
    # nothing to plan: an empty matrix gives an inactive plan whose calls run the ordinary path
    empty = Dev(torch, dev, np.zeros(11, np.int32), np.zeros(0, np.int32), np.zeros(0), 6)
    p0 = sblas.SpmmPlan(10, 6, empty.rowptr, empty.colidx, 64)
    assert not p0.info()["active"]
    C = torch.arange(10 * 64, dtype=torch.float64, device=dev)
    p0.spmm(empty.val, torch.ones(6 * 64, dtype=torch.float64, device=dev), 6, 64, 1.0, 3.0, C, 10, None)
    assert torch.equal(C, 3.0 * torch.arange(10 * 64, dtype=torch.float64, device=dev))


@pytest.mark.parametrize("n", [64, 256])
def test_planned_row_block_stages_its_column_range_without_the_range_pass(env, stage_range_env, n):
    """A method-2 row block: the plan holds the block's column range, so a planned call neither reads the column indices
    again nor classifies; the workspace is poisoned and B holds Inf outside the range."""
    sblas, oracle, torch, dev = env
    stage_range_env("1")
    K, r0, rows, band = 20000, 9000, 1500, 300
    rng = np.random.default_rng(77)
    lens = rng.integers(20, 90, rows)
    rp = np.zeros(rows + 1, np.int64)
    rp[1:] = np.cumsum(lens)
    ci = np.empty(rp[-1], np.int32)
    for r in range(rows):
        ci[rp[r]:rp[r + 1]] = np.sort(rng.choice(np.arange(r0 + r - band, r0 + r + band), lens[r], replace=False))
    v = rng.standard_normal(rp[-1])
    A = Dev(torch, dev, rp.astype(np.int32), ci, v, K)
    B = rng.standard_normal(K * n)
    Bm = B.reshape(n, K)
    cmin, cmax = int(ci.min()), int(ci.max())
    Bm[:, :cmin] = np.inf
    Bm[:, cmax + 1:] = -np.inf
    C0 = rng.standard_normal(rows * n)
    plan = sblas.SpmmPlan(rows, K, A.rowptr, A.colidx, n)
    assert plan.info()["stage_range"]
    Bd, Cd = torch.from_numpy(B).to(dev), torch.from_numpy(C0.copy()).to(dev)
    ws = torch.full((sblas.spmm_workspace_bytes(rows, K, len(ci), n) // 8,), float("nan"), dtype=torch.float64, device=dev)
    plan.spmm(A.val, Bd, K, n, 1.5, -0.5, Cd, rows, ws)
    torch.cuda.synchronize()
    Bo = B.copy().reshape(n, K)
    Bo[:, :cmin] = 0.0
    Bo[:, cmax + 1:] = 0.0
    ref = oracle.spmm(rows, K, n, *A.h, Bo.reshape(-1), C0.copy(), 1.5, -0.5)
    assert close(Cd.cpu().numpy(), ref)
    plan.destroy()


def test_planned_call_inside_a_hip_graph(env):
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows, n = 4000, 64
    rp, ci, v = synth.banded(rows, 150, 600)
    A = Dev(torch, dev, rp, ci, v, rows)
    rng = np.random.default_rng(11)
    B = torch.from_numpy(rng.standard_normal(rows * n)).to(dev)
    C = torch.zeros(rows * n, dtype=torch.float64, device=dev)
    ws = torch.empty(sblas.spmm_workspace_bytes(rows, rows, len(ci), n) // 8, dtype=torch.float64, device=dev)
    plan = sblas.SpmmPlan(rows, rows, A.rowptr, A.colidx, n)           # (creating a plan synchronises: outside the capture)
    plan.spmm(A.val, B, rows, n, 1.0, 0.0, C, rows, ws)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.spmm(A.val, B, rows, n, 2.0, 0.0, C, rows, ws)
    for rep in range(2):
        Bh = rng.standard_normal(rows * n)
        B.copy_(torch.from_numpy(Bh).to(dev))
        C.fill_(7.0)
        g.replay()
        torch.cuda.synchronize()
        assert close(C.cpu().numpy(), oracle.spmm(rows, rows, n, *A.h, Bh, np.zeros(rows * n), 2.0, 0.0)), rep
    plan.destroy()


# ---------------------------------------------------------------------------------------------------------
# ADVICE r2 / VERDICT r2 housekeeping
# ---------------------------------------------------------------------------------------------------------
def test_validate_switch_refuses_column_indices_outside_the_matrix(env):
    """The compute kernels trust the index arrays (a column index >= cols is an out-of-bounds read of the staging copy, as
    with the vendor libraries); SBLAS_VALIDATE=1 / sblas_hip_debug_validate_csr_i32 check them first."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows = 500
    rp, ci, v = synth.banded(rows, 20, 60)
    A = Dev(torch, dev, rp, ci, v, rows)
    assert sblas.validate_csr(rows, rows, A.rowptr, A.colidx)
    bad_ci = ci.copy()
    bad_ci[1234] = rows + 7
    Abad = Dev(torch, dev, rp, bad_ci, v, rows)
    assert not sblas.validate_csr(rows, rows, Abad.rowptr, Abad.colidx)
    bad_rp = rp.copy()
    bad_rp[100] = bad_rp[101] + 3                                      # a row pointer that runs backwards
    assert not sblas.validate_csr(rows, rows, torch.from_numpy(bad_rp).to(dev), A.colidx)
    B = torch.ones(rows * 64, dtype=torch.float64, device=dev)
    C = torch.zeros(rows * 64, dtype=torch.float64, device=dev)
    ws = torch.empty(sblas.spmm_workspace_bytes(rows, rows, len(ci), 64) // 8, dtype=torch.float64, device=dev)
    x, y = torch.ones(rows, dtype=torch.float64, device=dev), torch.zeros(rows, dtype=torch.float64, device=dev)
    for setter in _env_switch("SBLAS_VALIDATE"):
        setter("1")
        with pytest.raises(sblas.SblasError):
            sblas.spmm(rows, rows, Abad.rowptr, Abad.colidx, Abad.val, B, rows, 64, 1.0, 0.0, C, rows, ws)
        with pytest.raises(sblas.SblasError):
            sblas.spmv(rows, rows, Abad.rowptr, Abad.colidx, Abad.val, x, 1.0, 0.0, y)
        sblas.spmm(rows, rows, A.rowptr, A.colidx, A.val, B, rows, 64, 1.0, 0.0, C, rows, ws)     # a good matrix passes
        torch.cuda.synchronize()
    assert close(C.cpu().numpy(), oracle.spmm(rows, rows, 64, *A.h, np.ones(rows * 64), np.zeros(rows * 64), 1.0, 0.0))


def test_nonfinite_flag_of_range_staging_does_not_stick_to_later_column_chunks(env):
    """a row block whose call walks two column chunks through the AUTOMATIC range staging (cols >> rows; no
    switch set).  B holds a NaN in the first chunk's columns only (in a row of the range that no nonzero refers to): the
    first chunk's matrix-core panels fall back to the vector kernels, the second chunk's must not -- every staging pass
    has an epoch of its own.  Block-structured rows (85 % fill) so that the matrix cores are chosen at 128 columns."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows, K, n, off = 1000, 400000, 256, 200004          # (a multiple of 4: the 16 x 4 blocks stay aligned)
    rp, ci, v = synth.block_structured(rows, nnz_per_row=150, half_band=400, fill=0.85)
    ci = (ci + off).astype(np.int32)
    A = Dev(torch, dev, rp, ci, v, K)
    lo, hi = int(ci.min()), int(ci.max())
    # a row of B that the staging pass copies (it walks whole 32-row tiles of the range) but no nonzero refers to
    free = lo - 1 if lo % 32 else hi + 1
    assert (free // 32 == lo // 32 or free // 32 == hi // 32) and not (lo <= free <= hi)
    rng = np.random.default_rng(3)
    B = torch.from_numpy(rng.standard_normal(K * n)).to(dev)
    B.view(n, K)[5, free] = float("nan")                               # column 5: the first 128-column chunk
    C = torch.zeros(rows * n, dtype=torch.float64, device=dev)
    for setter in _env_switch("SBLAS_SPMM_MAX_BT_BYTES"):
        setter(str(450 * 1000 * 1000))                                 # a 128-column staging copy fits, 256 columns do not
        ws = torch.full((sblas.spmm_workspace_bytes(rows, K, len(ci), n) // 8,), float("nan"), dtype=torch.float64, device=dev)
        assert ws.numel() * 8 < 450 * 1000 * 1000
        sblas.panel_census()
        sblas.spmm(rows, K, A.rowptr, A.colidx, A.val, B, K, n, 1.0, 0.0, C, rows, ws)
        torch.cuda.synchronize()
        census = sblas.panel_census()
    Bt = ws[: (K + 1) * 128].view(K + 1, 128)
    assert bool(torch.isnan(Bt[: lo - 32]).all())                      # the automatic rule took the range staging
    assert census["mfma"] > 0, census                                  # ... and the second chunk kept its matrix-core panels
    assert census["windowed"] + census["direct"] > 0, census           # the first chunk's went to the vector kernels
    Bh = B.cpu().numpy()
    got = C.cpu().numpy()
    assert np.isfinite(got).all()
    r0 = 500
    ref = np.zeros(rows * n)
    oracle.spmm_rows(r0, r0 + 64, rows, K, n, *A.h, np.nan_to_num(Bh), ref, 1.0, 0.0)
    assert close(got.reshape(n, rows)[:, r0:r0 + 64], ref.reshape(n, rows)[:, r0:r0 + 64])


def test_forced_matrix_core_variant_keeps_panels_the_kernel_can_take(env):
    """around 33 000 rows the panel plan picks 132- / 144-row panels (three groups per wave), which the
    matrix-core kernel (16 rows per wave, eight waves) leaves alone; SBLAS_SPMM_VARIANT=mfma now plans two groups."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows, n = 33000, 64
    rp, ci, v = synth.block_structured(rows, nnz_per_row=60, half_band=300, fill=0.6)
    A = Dev(torch, dev, rp, ci, v, rows)
    rng = np.random.default_rng(1)
    Bh = rng.standard_normal(rows * n)
    B = torch.from_numpy(Bh).to(dev)
    C = torch.zeros(rows * n, dtype=torch.float64, device=dev)
    ws = torch.empty(sblas.spmm_workspace_bytes(rows, rows, len(ci), n) // 8, dtype=torch.float64, device=dev)
    for setter in _env_switch("SBLAS_SPMM_VARIANT"):
        setter("mfma")
        sblas.panel_census()
        sblas.spmm(rows, rows, A.rowptr, A.colidx, A.val, B, rows, n, 1.0, 0.0, C, rows, ws)
        torch.cuda.synchronize()
        census = sblas.panel_census()
    assert census["mfma"] > 0 and census["windowed"] == 0, census
    ref = np.zeros(rows * n)
    oracle.spmm_rows(1000, 1064, rows, rows, n, *A.h, Bh, ref, 1.0, 0.0)
    assert close(C.cpu().numpy().reshape(n, rows)[:, 1000:1064], ref.reshape(n, rows)[:, 1000:1064])


@pytest.mark.parametrize("n", [128, 200])
def test_direct_kernel_vote_at_128_columns(env, n):
    """Round 3: from 128 staged columns on the call's direct panels go to the four-rows-per-wave kernel unless half of
    them show column runs (or row lengths far apart) -- then a row per wave on 128-column tiles keeps them; rows in groups
    of three equal patterns still go to the row-merging kernel.  The plan reports the verdict; planned and unplanned calls
    agree bit for bit and with the oracle either way."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rng = np.random.default_rng(5)
    rows = 3000
    # (a) banded-random rows, 40 per row over +-1200 columns: no runs -> four rows per wave
    a = synth.banded(rows, 40, 1200)
    # (b) clusters of three consecutive columns at scattered offsets, every row its own choice (Queen-like): runs -> row per wave
    b = synth.queen_like(rows, half_band=1200)
    # (c) rows of very different lengths (one in eight is ten times the others): lengths far apart -> row per wave
    lens = np.where(np.arange(rows) % 8 == 0, 400, 40)
    rp = np.zeros(rows + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    ci = np.concatenate([np.sort(rng.choice(rows, l, replace=False)) for l in lens]).astype(np.int32)
    c = (rp.astype(np.int32), ci, rng.standard_normal(len(ci)))
    # (d) grid-structured rows, three unknowns per node: row merging
    d = synth.queen_like_grid(rows, half_band=400)
    for name, (rp_, ci_, v_), want in (("banded", a, "four_rows"), ("clusters", b, None), ("skewed", c, None), ("grid", d, "merge")):
        m = len(rp_) - 1
        A = Dev(torch, dev, rp_, ci_, v_, m)
        Bh, C0 = rng.standard_normal(m * n), rng.standard_normal(m * n)
        B = torch.from_numpy(Bh).to(dev)
        ws = torch.empty(sblas.spmm_workspace_bytes(m, m, len(ci_), n) // 8, dtype=torch.float64, device=dev)
        plan = sblas.SpmmPlan(m, m, A.rowptr, A.colidx, n)
        info = plan.info()
        assert info["active"] and info["direct"] > 0, (name, info)
        assert info["four_rows"] == (want == "four_rows") and info["merge"] == (want == "merge"), (name, info)
        Cu, Cp = torch.from_numpy(C0.copy()).to(dev), torch.from_numpy(C0.copy()).to(dev)
        sblas.spmm(m, m, A.rowptr, A.colidx, A.val, B, m, n, 0.5, 2.0, Cu, m, ws)
        plan.spmm(A.val, B, m, n, 0.5, 2.0, Cp, m, ws)
        torch.cuda.synchronize()
        assert torch.equal(Cu, Cp), name
        assert close(Cp.cpu().numpy(), oracle.spmm(m, m, n, *A.h, Bh, C0.copy(), 0.5, 2.0)), name
        plan.destroy()
