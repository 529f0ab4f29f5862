"""The value / index types besides <int32, fp64>: the reference's templates take float or double values and 32- or
64-bit indices (utility.h:302-316 maps them onto cuSPARSE; spmm.h:109-118, spmv.h:64-77).  Every call goes through
the typed C-ABI entry points (sblas_hip_spmm_csr, sblas_hip_spmv_csr, ...); the checker is the oracle's restatement
of sblas_spmm_csr_cpu<IdxType, DataType> / sblas_spmv_csr_cpu<IdxType, DataType> in the same types.

Tolerances: fp64 as everywhere (1e-10 relative); fp32: 1e-4 relative + 1e-4 absolute -- sums of up to a few hundred
fp32 terms of O(1), fused multiply-add on the GPU against multiply-then-add in the oracle.  "parity unpinned": the
reference holds no fp32 / int64 outputs (its drivers instantiate <int, double> only)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ASH85, ROOT

pytestmark = pytest.mark.gpu

TYPES = [("f32", "i32"), ("f64", "i64"), ("f32", "i64"), ("f64", "i32")]
NP = {"f32": np.float32, "f64": np.float64, "i32": np.int32, "i64": np.int64}


def close(got, ref):
    if ref.dtype == np.float32:
        return np.allclose(got, ref, rtol=1e-4, atol=1e-4)
    return np.allclose(got, ref, rtol=1e-10, atol=1e-12)


@pytest.fixture(scope="module")
def env(sblas, oracle, cuda):
    import torch
    return sblas, oracle, torch, cuda


def _up(torch, dev, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _spmm(sblas, torch, dev, rows, cols, rp, ci, v, B, ldb, n, alpha, beta, C0, ldc, c_offset=0):
    drp, dci, dv = _up(torch, dev, rp), _up(torch, dev, ci), _up(torch, dev, v)
    dB, dC = _up(torch, dev, B), _up(torch, dev, C0.copy())
    nbytes = sblas.spmm_typed_workspace_bytes(dv.dtype, drp.dtype, rows, cols, len(ci), n)
    ws = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device=dev) if nbytes else None     # all-ones bytes = NaN
    sblas.spmm_typed(rows, cols, drp, dci, dv, dB, ldb, n, alpha, beta, dC, ldc, ws, c_offset=c_offset)
    torch.cuda.synchronize()
    return dC.cpu().numpy()


@pytest.mark.parametrize("vt,it", TYPES)
@pytest.mark.parametrize("n", [1, 7, 64, 65, 130])
def test_typed_spmm_matches_the_typed_host_loop(env, vt, it, n):
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    M, K = 150, 97
    rp, ci, v = synth.random_csr(M, K, 9, seed=n, empty_every=11, long_row=(5, 333))
    rp, ci, v = rp.astype(NP[it]), ci.astype(NP[it]), v.astype(NP[vt])
    rng = np.random.default_rng(n)
    B, C0 = rng.standard_normal(K * n).astype(NP[vt]), rng.standard_normal(M * n).astype(NP[vt])
    got = _spmm(sblas, torch, dev, M, K, rp, ci, v, B, K, n, 1.5, -0.5, C0, M)
    assert got.dtype == NP[vt]
    assert close(got, oracle.spmm_typed(M, K, n, rp, ci, v, B, C0.copy(), 1.5, -0.5))


@pytest.mark.parametrize("vt,it", TYPES[:3])
def test_typed_spmm_leading_dimensions_beta_zero_and_row_blocks(env, vt, it):
    """ldb > cols and ldc > rows with untouched padding; beta = 0 over a NaN C; a method-2 style row block written at
    an offset into a taller C (re-based row pointers); a matrix without nonzeros scales C and needs no workspace."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    M, K, n = 77, 300, 70
    rp, ci, v = synth.random_csr(M, K, 12, seed=2, empty_every=5)
    rp, ci, v = rp.astype(NP[it]), ci.astype(NP[it]), v.astype(NP[vt])
    rng = np.random.default_rng(8)
    ldb, ldc = K + 5, M + 3
    Bp = rng.standard_normal(ldb * n).astype(NP[vt])
    Cp = rng.standard_normal(ldc * n).astype(NP[vt])
    got = _spmm(sblas, torch, dev, M, K, rp, ci, v, Bp, ldb, n, 2.0, 0.25, Cp, ldc)
    Bd = np.ascontiguousarray(Bp.reshape(n, ldb)[:, :K]).reshape(-1)
    Cd = np.ascontiguousarray(Cp.reshape(n, ldc)[:, :M]).reshape(-1)
    ref = oracle.spmm_typed(M, K, n, rp, ci, v, Bd, Cd.copy(), 2.0, 0.25)
    assert close(got.reshape(n, ldc)[:, :M], ref.reshape(n, M))
    assert np.array_equal(got.reshape(n, ldc)[:, M:], Cp.reshape(n, ldc)[:, M:])            # padding rows untouched
    # beta = 0: C is not read
    Cn = np.full(M * n, np.nan, NP[vt])
    got = _spmm(sblas, torch, dev, M, K, rp, ci, v, Bd, K, n, 1.0, 0.0, Cn, M)
    assert close(got, oracle.spmm_typed(M, K, n, rp, ci, v, Bd, np.zeros(M * n, NP[vt]), 1.0, 0.0))
    # row block [20, 60) at its offset in the full-height C
    a, b = 20, 60
    sub = (rp[a:b + 1] - rp[a]).astype(NP[it])
    Cfull = rng.standard_normal(M * n).astype(NP[vt])
    got = _spmm(sblas, torch, dev, b - a, K, sub, ci[rp[a]:rp[b]], v[rp[a]:rp[b]], Bd, K, n, 1.0, 1.0, Cfull, M, c_offset=a)
    part = oracle.spmm_typed(b - a, K, n, sub, ci[rp[a]:rp[b]], v[rp[a]:rp[b]], Bd, np.zeros((b - a) * n, NP[vt]), 1.0, 0.0)
    want = Cfull.copy().reshape(n, M)
    want[:, a:b] += part.reshape(n, b - a)
    assert close(got, want.reshape(-1))
    # no nonzeros: C = beta * C, NULL workspace
    rp0 = np.zeros(M + 1, NP[it])
    got = _spmm(sblas, torch, dev, M, K, rp0, ci[:0], v[:0], Bd, K, n, 1.0, 0.5, Cd, M)
    assert close(got, (Cd * NP[vt](0.5)).astype(NP[vt]))


@pytest.mark.parametrize("vt,it", TYPES)
@pytest.mark.parametrize("avg", [1, 7, 40, 400])
def test_typed_spmv_matches_the_typed_host_loop(env, vt, it, avg):
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    M = K = 2000
    rp, ci, v = synth.random_csr(M, K, avg, seed=avg, empty_every=13, long_row=(17, min(K, 1500)))
    rp, ci, v = rp.astype(NP[it]), ci.astype(NP[it]), v.astype(NP[vt])
    rng = np.random.default_rng(avg)
    x, y0 = rng.standard_normal(K).astype(NP[vt]), rng.standard_normal(M).astype(NP[vt])
    drp, dci, dv = _up(torch, dev, rp), _up(torch, dev, ci), _up(torch, dev, v)
    dx, dy = _up(torch, dev, x), _up(torch, dev, y0.copy())
    sblas.spmv_typed(M, K, drp, dci, dv, dx, 3.0, 4.0, dy)
    assert close(dy.cpu().numpy(), oracle.spmv_typed(M, rp, ci, v, x, y0.copy(), 3.0, 4.0))
    # beta = 0 over NaN, and a row block at an offset
    dy = torch.full((M,), float("nan"), dtype=dv.dtype, device=dev)
    sblas.spmv_typed(M, K, drp, dci, dv, dx, 1.0, 0.0, dy)
    assert close(dy.cpu().numpy(), oracle.spmv_typed(M, rp, ci, v, x, np.zeros(M, NP[vt]), 1.0, 0.0))
    a, b = 500, 1300
    sub = _up(torch, dev, (rp[a:b + 1] - rp[a]).astype(NP[it]))
    dy = _up(torch, dev, y0.copy())
    sblas.spmv_typed(b - a, K, sub, dci[rp[a]:rp[b]], dv[rp[a]:rp[b]], dx, 1.0, 1.0, dy, y_offset=a)
    want = y0.copy()
    want[a:b] += oracle.spmv_typed(M, rp, ci, v, x, np.zeros(M, NP[vt]), 1.0, 0.0)[a:b]
    assert close(dy.cpu().numpy(), want)


@pytest.mark.parametrize("n", [1, 255, 100001])
def test_typed_axpby_fp32(env, n):
    sblas, oracle, torch, dev = env
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    dx, dy = _up(torch, dev, x), _up(torch, dev, y.copy())
    sblas.axpby_typed(n, 3.0, dx, 4.0, dy)
    assert np.allclose(dy.cpu().numpy(), y * np.float32(4.0) + x * np.float32(3.0), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("vt", ["f32", "f64"])
@pytest.mark.parametrize("g", [2, 4])
def test_typed_method2_merges_on_folded_ranks(env, vt, g):
    """Method 2 in the value type on g ranks folded onto the one device: per-rank SpMM of the nnz row blocks, then
    both merges (packed row blocks + scatter; zero-filled M x N + all-reduce + axpby) through the typed entry points."""
    sblas, oracle, torch, dev = env
    from sblas_amd import synth
    M, K, N = 400, 350, 40
    rp, ci, v = synth.random_csr(M, K, 10, seed=g, long_row=(100, 300))
    v = v.astype(NP[vt])
    rng = np.random.default_rng(g)
    Bh, C0 = rng.standard_normal(K * N).astype(NP[vt]), rng.standard_normal(M * N).astype(NP[vt])
    ref = oracle.spmm_typed(M, K, N, rp, ci, v, Bh, C0.copy(), 3.0, 4.0)
    comm = sblas.comm_get([0] * g)
    streams = [torch.cuda.Stream(device=dev) for _ in range(g)]
    dB = _up(torch, dev, Bh)
    parts = [sblas.partition_nnz(rp, g, q) for q in range(g)]
    starts = [p["start_row"] for p in parts]
    nrows = [len(p["rowptr"]) - 1 for p in parts]
    torch.cuda.synchronize()
    for merge in ("rowblocks", "allreduce"):
        Cs = [_up(torch, dev, C0.copy()) for _ in range(g)]
        partial = []
        torch.cuda.synchronize()
        for q in range(g):
            lo, k, m_i = parts[q]["first_nnz"], parts[q]["nnz"], nrows[q]
            drp, dci, dv = _up(torch, dev, parts[q]["rowptr"]), _up(torch, dev, ci[lo:lo + k]), _up(torch, dev, v[lo:lo + k])
            ws = torch.empty(sblas.spmm_typed_workspace_bytes(dv.dtype, drp.dtype, m_i, K, k, N), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            with torch.cuda.stream(streams[q]):
                if merge == "rowblocks":
                    blk = torch.full((m_i * N,), 7.0, dtype=dv.dtype, device=dev)
                    sblas.spmm_typed(m_i, K, drp, dci, dv, dB, K, N, 1.0, 0.0, blk, m_i, ws, stream=streams[q])
                else:
                    blk = torch.zeros(M * N, dtype=dv.dtype, device=dev)
                    sblas.spmm_typed(m_i, K, drp, dci, dv, dB, K, N, 1.0, 1.0, blk, M, ws, stream=streams[q], c_offset=starts[q])
            partial.append(blk)
            torch.cuda.synchronize()       # the per-rank temporaries above go out of scope
        if merge == "rowblocks":
            sblas.merge_rowblocks_typed(comm, M, N, starts, nrows, partial, None, 3.0, 4.0, Cs, M, streams)
        else:
            sblas.allreduce_sum_typed(comm, partial, streams, M * N)
            for q in range(g):
                sblas.axpby_typed(M * N, 3.0, partial[q], 4.0, Cs[q], stream=streams[q])
        torch.cuda.synchronize()
        for q in range(g):
            assert close(Cs[q].cpu().numpy(), ref), (merge, q)


def test_typed_driver_runs_every_instantiation(sblas):
    """bin/typed_test: sblas_spmm_csr_v1 / _v2 (both merges) / sblas_spmv_csr_v1 instantiated for <int, float>,
    <int64_t, double>, <int64_t, float> and <int, double> on ash85 with four (folded) GPUs, each against the host loop
    of the same types."""
    exe = os.path.join(ROOT, "s-blas_amd", "bin", "typed_test")
    p = subprocess.run([exe, ASH85, "4", "100"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "typed_test: PASS" in p.stdout and "FAIL" not in p.stdout
