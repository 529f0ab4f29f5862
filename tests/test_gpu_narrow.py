"""GPU parity tests of the narrow dense widths (n <= 32: the column blocks method 1 hands a GPU when N = 64 is split over
2 / 4 / 8 of them, matrix.h:554-568): the LDS-tiled lane-per-entry kernel (spmm_lanes_kernel), its per-panel fallback,
the direct kernels on the panels it does not take, and the 8- / 16- / 32-column staging copies -- against the CPU oracle
through the C ABI.  fp64 within 1e-10 relative (north_star)."""
import os

import numpy as np
import pytest

from test_gpu_parity import Dev, close, gpu_spmm, _env_switch, SPMM_VARIANTS

pytestmark = pytest.mark.gpu

NARROW_N = [1, 5, 8, 9, 12, 16, 17, 24, 32]


@pytest.fixture(scope="module")
def env(sblas, oracle, cuda):
    import torch
    assert sblas.lib().sblas_hip_device_count() >= 1
    return sblas, oracle, torch, cuda


@pytest.fixture
def variant_env():
    yield from _env_switch("SBLAS_SPMM_VARIANT")


@pytest.fixture
def panel_rows_env():
    yield from _env_switch("SBLAS_SPMM_PANEL_ROWS")


@pytest.fixture
def tune_env():
    yield from _env_switch("SBLAS_TUNE")


def test_narrow_staged_widths(env):
    """8 / 16 / 32 staged columns for n <= 8 / 16 / 32 (the widths method 1 produces), 64 above."""
    sblas = env[0]
    ld = lambda n: int(sblas.lib().sblas_hip_spmm_ldbt(n))
    assert [ld(n) for n in (1, 8, 9, 16, 17, 32, 33, 64, 65)] == [8, 8, 16, 16, 32, 32, 64, 64, 128]


@pytest.mark.parametrize("variant", SPMM_VARIANTS)
@pytest.mark.parametrize("n", NARROW_N)
def test_narrow_banded_every_width_and_selection(env, variant_env, variant, n):
    """Banded ascending rows, dense enough over their span for the LDS-tiled kernel; several 256-row tiles per panel,
    K not a multiple of the tile, n below the staged width (zero padding), alpha / beta non-trivial and beta = 0."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    variant_env(variant)
    rows = 1100
    rp, ci, v = synth.banded(rows, 70, 330, seed=n)                  # (the narrow kernel takes panels from 56 / 20 / 16 nonzeros per row on)
    A = Dev(torch, dev, rp, ci, v, rows)
    rng = np.random.default_rng(n)
    B, C0 = rng.standard_normal(rows * n), rng.standard_normal(rows * n)
    for alpha, beta in ((1.25, -0.5), (1.0, 0.0)):
        sblas.panel_census()
        got = gpu_spmm(sblas, torch, dev, A, B, rows, n, alpha, beta, C0, rows)
        census = sblas.panel_census()
        ref = oracle.spmm(rows, rows, n, *A.h, B, C0.copy(), alpha, beta)
        assert close(got, ref), (variant, n, alpha, beta, np.abs(got - ref).max())
        if variant == "auto":
            assert census["windowed"] > 0 and census["direct"] == 0 and census["fallback"] == 0, census


@pytest.mark.parametrize("n", [8, 16, 32])
@pytest.mark.parametrize("damage", ["all_descending", "one_row_shuffled", "first_col_not_min", "duplicates"])
def test_narrow_fallback_on_unsorted_rows(env, n, damage):
    """The LDS-tiled kernel expects ascending columns but never depends on it: panels that break the expectation are
    detected in the kernel and recomputed from global memory before C is written."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows = 600
    rp, ci, v = synth.banded(rows, 70, 120 if n == 16 else 400)      # rows inside one 256-row tile / across several
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
        for r in range(0, rows, 7):
            a, b = rp[r], rp[r + 1] - 1
            ci[a], ci[b] = ci[b], ci[a]
            v[a], v[b] = v[b], v[a]
    else:
        for r in range(0, rows, 3):
            ci[rp[r] + 1] = ci[rp[r]]
            ci[rp[r + 1] - 1] = ci[rp[r + 1] - 2]
    A = Dev(torch, dev, rp, ci, v, rows)
    B, C0 = rng.standard_normal(rows * n), rng.standard_normal(rows * n)
    got = gpu_spmm(sblas, torch, dev, A, B, rows, n, 1.0, 1.0, C0, rows)
    ref = oracle.spmm(rows, rows, n, *A.h, B, C0.copy(), 1.0, 1.0)
    assert close(got, ref), (n, damage, np.abs(got - ref).max())


@pytest.mark.parametrize("n", [7, 16, 29])
def test_narrow_mixed_panels_row_block_and_padding(env, n):
    """Dense-band panels next to sparse wide-span panels (the direct kernel computes those), empty rows, a row of 900
    entries (several windows per tile visit), then a method-2 style row block: re-based row pointers, C written at a
    row offset into a taller C (ldc > rows), ldb > K; the rows of C outside the block stay untouched."""
    sblas, oracle, torch, dev = env
    rows, K = 512, 3000
    rng = np.random.default_rng(21 + n)
    lens = np.where(np.arange(rows) % 128 < 64, 200, 3).astype(np.int64)
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
    sblas.panel_census()
    got = gpu_spmm(sblas, torch, dev, A, B, K, n, 2.0, 0.5, C0, rows)
    census = sblas.panel_census()
    ref = oracle.spmm(rows, K, n, *A.h, B, C0.copy(), 2.0, 0.5)
    assert close(got, ref)
    assert census["windowed"] > 0 and census["direct"] > 0, census
    ldb = K + 13
    Bp = rng.standard_normal(ldb * n)
    sub = rp[100:401] - rp[100]
    As = Dev(torch, dev, sub.astype(np.int32), ci[rp[100]:rp[400]], v[rp[100]:rp[400]], K)
    Cbig = rng.standard_normal(rows * n)
    Bd = torch.from_numpy(Bp).to(dev)
    Cd = torch.from_numpy(Cbig.copy()).to(dev)
    ws = torch.empty(sblas.spmm_workspace_bytes(300, K, 1, n) // 8, dtype=torch.float64, device=dev)
    sblas.spmm(300, K, As.rowptr, As.colidx, As.val, Bd, ldb, n, 1.0, 1.0, Cd, rows, ws, c_offset=100)
    want = Cbig.copy().reshape(n, rows)
    part = np.zeros(300 * n)
    oracle.spmm(300, K, n, *As.h, np.ascontiguousarray(Bp.reshape(n, ldb)[:, :K]).reshape(-1), part, 1.0, 0.0)
    want[:, 100:400] += part.reshape(n, 300)
    got = Cd.cpu().numpy().reshape(n, rows)
    assert close(got, want)
    assert (got[:, :100] == Cbig.reshape(n, rows)[:, :100]).all() and (got[:, 400:] == Cbig.reshape(n, rows)[:, 400:]).all()


@pytest.mark.parametrize("n,setting", [(8, "48,1"), (8, "96,2"), (8, "144,3"), (16, "48,1"), (16, "96,2"), (16, "144,3"), (32, "48,1"),
                                       (32, "96,2")])
def test_narrow_census_and_every_panel_height(env, panel_rows_env, n, setting):
    """Every instantiated (width, groups per wave) pair; the census tells an LDS-tiled run from a fallback or a direct
    one ([windowed, direct, fallback])."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    panel_rows_env(setting)
    pr = int(setting.split(",")[0])

    def run(rp, ci, v, rows, cols):
        A = Dev(torch, dev, rp, ci, v, cols)
        rng = np.random.default_rng(1)
        B, C0 = rng.standard_normal(cols * n), rng.standard_normal(rows * n)
        sblas.panel_stats()
        got = gpu_spmm(sblas, torch, dev, A, B, cols, n, 1.0, 1.0, C0, rows)
        st = sblas.panel_stats()
        assert close(got, oracle.spmm(rows, cols, n, *A.h, B, C0.copy(), 1.0, 1.0))
        return st

    rows = 10 * pr - 7                                                    # the last panel is short
    rp, ci, v = synth.banded(rows, 70, 400)                               # rows span three or four 256-row tiles
    assert run(rp, ci, v, rows, rows) == (10, 0, 0)                       # every panel through LDS
    ci2, v2 = ci.copy(), v.copy()
    r = 2 * pr + 4                                                        # (far from the matrix edges: a full-width row)
    ci2[rp[r]:rp[r + 1]] = ci2[rp[r]:rp[r + 1]][::-1]                     # one descending row in panel 2
    v2[rp[r]:rp[r + 1]] = v2[rp[r]:rp[r + 1]][::-1]
    assert run(rp, ci2, v2, rows, rows) == (9, 0, 1)                      # that panel recomputed, the others not
    rp3, ci3, v3 = synth.random_csr(rows, 5000, 60, seed=3, sorted_rows=True)
    assert run(rp3, ci3, v3, rows, 5000) == (0, 10, 0)                    # too sparse over its span: direct kernel
    rp5, ci5, v5 = synth.random_csr(rows, 5000, 4, seed=3, sorted_rows=True)
    assert run(rp5, ci5, v5, rows, 5000) == (0, 0, 0)                     # rows far below the bar: the call does not classify at all
    rp4, ci4, v4 = synth.banded(rows, 500, 400)                           # rows of 500: several windows per tile visit
    assert run(rp4, ci4, v4, rows, rows) == (10, 0, 0)


@pytest.mark.parametrize("n,copies,tune", [(8, 2, "2,0,0,0"), (8, 4, "4,0,0,0"), (16, 2, "2,0,0,0"), (32, 1, "0,0,0,1")])
def test_narrow_tile_rows_stored_more_than_once(env, tune_env, n, copies, tune):
    """SBLAS_TUNE=<copies>: the LDS tile holds every Bt row `copies` times over (bank-conflict experiment, kept
    instantiated); SBLAS_TUNE=0,0,0,1: 32 columns with a lane per entry instead of two lanes per entry: same results."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    tune_env(tune)
    rows = 900
    rp, ci, v = synth.banded(rows, 70, 300)
    A = Dev(torch, dev, rp, ci, v, rows)
    rng = np.random.default_rng(copies)
    B, C0 = rng.standard_normal(rows * n), rng.standard_normal(rows * n)
    got = gpu_spmm(sblas, torch, dev, A, B, rows, n, -1.5, 2.0, C0, rows)
    assert close(got, oracle.spmm(rows, rows, n, *A.h, B, C0.copy(), -1.5, 2.0))


@pytest.mark.parametrize("n", [8, 16, 32])
def test_narrow_nonfinite_b_rows_not_referenced_stay_out(env, variant_env, n):
    """Masked lanes read the all-zero row, never a real row of B: an Inf / NaN in a row of B that no nonzero refers
    to must not reach C (0 * Inf = NaN) -- in the LDS tiles (the tile holds such rows) and in the direct kernels."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows = 700
    rp, ci, v = synth.banded(rows, 64, 200)
    ci = ci.copy()
    ci[ci == 0] = 1
    ci[ci % 5 == 3] += 1                              # nobody refers to columns = 3 (mod 5) ...
    ci = np.minimum(ci, rows - 1)
    for r in range(rows):
        ci[rp[r]:rp[r + 1]] = np.sort(ci[rp[r]:rp[r + 1]])
    used = np.zeros(rows, bool)
    used[ci] = True
    A = Dev(torch, dev, rp, ci, v, rows)
    rng = np.random.default_rng(4)
    B = rng.standard_normal(rows * n).reshape(n, rows)
    B[:, ~used] = np.inf
    B[:, 0] = np.nan
    B = np.ascontiguousarray(B).reshape(-1)
    C0 = rng.standard_normal(rows * n)
    for variant in ("auto", "dpp", "lanes"):
        variant_env(variant)
        got = gpu_spmm(sblas, torch, dev, A, B, rows, n, 1.0, 1.0, C0, rows)
        ref = oracle.spmm(rows, rows, n, *A.h, B, C0.copy(), 1.0, 1.0)
        assert np.isfinite(ref).all() and np.isfinite(got).all() and close(got, ref), (variant, n)


@pytest.mark.parametrize("g", [2, 4, 8])
def test_method1_strong_column_blocks_at_bench_structure(env, g):
    """Method 1 on g GPUs at N = 64 (matrix.h:554-568): GPU i multiplies the full A by ceil(64 / g) = 32 / 16 / 8
    columns.  The bench workload's structure (399 per row, band +-2000) at 4 % of its rows; the blocks together must
    equal the verifier on all 64 columns."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    rows, (rp, ci, v) = synth.nd24k_like(scale=0.04)
    A = Dev(torch, dev, rp, ci, v, rows)
    N = 64
    B = oracle.rand0to1(rows * N)
    C = np.ones(rows * N)
    for i in range(g):
        off, dim = sblas.partition_dense(N, g, i)
        assert dim == N // g
        sblas.panel_census()
        blk = gpu_spmm(sblas, torch, dev, A, B[off * rows:(off + dim) * rows].copy(), rows, dim, 3.0, 4.0,
                       C[off * rows:(off + dim) * rows].copy(), rows)
        assert sblas.panel_census()["windowed"] > 0
        C[off * rows:(off + dim) * rows] = blk
    ref = oracle.spmm(rows, rows, N, *A.h, B, np.ones(rows * N), 3.0, 4.0)
    assert close(C, ref)
