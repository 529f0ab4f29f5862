"""Runs in a process of its own (tests/test_gpu_comm_stub.py starts it): comm.hip's EXCHANGE branches -- persistent
communicator, grouped ncclSend / ncclRecv of the packed row blocks into the gather buffers, grouped ncclAllReduce --
on ONE GPU, against tests/rccl_stub's stand-in for librccl (SBLAS_RCCL_LIB) with the ranks' equal device ids sent
down the distinct-device path (SBLAS_COMM_FORCE_EXCHANGE=1).  The environment must be set before the library resolves
RCCL, hence the separate process.  Cases: BASELINE config 4 (method 2, g = 4, N = 128, reduced rows) with both merges,
g = 3 with an empty block, the SpMV merge (N = 1), and the fp32 merges (RCCL_FLOAT32, byte offsets of the receives)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "s-blas_amd", "python"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

STUB = os.path.join(ROOT, "tests", "rccl_stub", "librccl_stub.so")
os.environ["SBLAS_RCCL_LIB"] = STUB
os.environ["SBLAS_COMM_FORCE_EXCHANGE"] = "1"

import numpy as np
import torch

import oracle_py as oracle
import sblas_amd as sblas
from sblas_amd import synth
from test_gpu_parity import Dev, _config4_run, close


def stub_stats(reset=True):
    lib = ctypes.CDLL(STUB)                      # the same handle comm.hip dlopen'ed (same path)
    out = (ctypes.c_longlong * 4)()
    lib.rccl_stub_stats(out, 1 if reset else 0)
    return dict(sends=out[0], allreduces=out[1], bytes=out[2], groups=out[3])


def main():
    assert torch.cuda.is_available(), "needs a GPU"
    sblas.lib()
    dev = 0
    rows, (rp, ci, v) = synth.nd24k_like(scale=0.05)
    N = 128
    # --- config 4, both merges, through the exchange branch
    stub_stats()
    _config4_run(sblas, oracle, torch, [dev] * 4, rows, rp, ci, v, N, "rowblocks")
    st = stub_stats()
    assert st["sends"] == 4 * 3 and st["allreduces"] == 0 and st["groups"] == 1, st
    assert st["bytes"] == 3 * rows * N * 8 + 3 * 3 * N * 8 or st["bytes"] >= 3 * rows * N * 8, st   # every block to 3 peers (+ cut rows)
    _config4_run(sblas, oracle, torch, [dev] * 4, rows, rp, ci, v, N, "allreduce")
    st = stub_stats()
    assert st["sends"] == 0 and st["allreduces"] == 1 and st["groups"] == 1, st
    # --- three ranks, the middle one without rows (an empty block must neither send nor be waited for)
    M, K, n = 300, 300, 16
    rng = np.random.default_rng(5)
    rp3, ci3, v3 = synth.banded(M, 20, 40)
    starts, nrows = [0, 150, 150], [150, 0, 150]
    Bh = rng.standard_normal(K * n)
    C0 = rng.standard_normal(M * n)
    comm = sblas.comm_get([dev] * 3)
    td = torch.device("cuda", dev)
    streams = [torch.cuda.Stream(device=td) for _ in range(3)]
    parts, gath, Cs = [], [], []
    B = torch.from_numpy(Bh).to(td)
    for q in range(3):
        m = nrows[q]
        part = torch.full((max(m * n, 1),), 7.0, dtype=torch.float64, device=td)
        if m:
            sub = (rp3[starts[q]:starts[q] + m + 1] - rp3[starts[q]]).astype(np.int32)
            lo, hi = rp3[starts[q]], rp3[starts[q] + m]
            A = Dev(torch, td, sub, ci3[lo:hi], v3[lo:hi], K)
            ws = torch.empty(max(sblas.spmm_workspace_bytes(m, K, hi - lo, n) // 8, 1), dtype=torch.float64, device=td)
            torch.cuda.synchronize()
            sblas.spmm(m, K, A.rowptr, A.colidx, A.val, B, K, n, 1.0, 0.0, part, m, ws, stream=streams[q])
        parts.append(part)
        gath.append(torch.empty(M * n, dtype=torch.float64, device=td))
        Cs.append(torch.from_numpy(C0.copy()).to(td))
    stub_stats()
    sblas.merge_rowblocks(comm, M, n, starts, nrows, parts, gath, 1.5, -0.5, Cs, M, streams)
    torch.cuda.synchronize()
    st = stub_stats()
    assert st["sends"] == 4, st                   # ranks 0 and 2 each to two peers; the empty rank sends nothing
    ref = oracle.spmm(M, K, n, rp3, ci3, v3, Bh, C0.copy(), 1.5, -0.5)
    for q in range(3):
        assert close(Cs[q].cpu().numpy(), ref), q
    # --- SpMV merge (N = 1), g = 4
    g = 4
    comm = sblas.comm_get([dev] * g)
    xh = np.random.default_rng(2).standard_normal(rows)
    yh = np.random.default_rng(3).standard_normal(rows)
    parts, ys, streams, starts, nrows = [], [], [], [], []
    x = torch.from_numpy(xh).to(td)
    for q in range(g):
        d = sblas.partition_nnz(rp, g, q)
        lo, k = d["first_nnz"], d["nnz"]
        Ai = Dev(torch, td, d["rowptr"], ci[lo:lo + k], v[lo:lo + k], rows)
        m_i = len(d["rowptr"]) - 1
        st_ = torch.cuda.Stream(device=td)
        yb = torch.zeros(max(m_i, 1), dtype=torch.float64, device=td)
        torch.cuda.synchronize()
        sblas.spmv(m_i, rows, Ai.rowptr, Ai.colidx, Ai.val, x, 1.0, 0.0, yb, stream=st_)
        parts.append(yb); streams.append(st_); starts.append(d["start_row"]); nrows.append(m_i)
        ys.append(torch.from_numpy(yh.copy()).to(td))
    gath = [torch.empty(max(sum(nrows), 1), dtype=torch.float64, device=td) for q in range(g)]
    stub_stats()
    sblas.merge_rowblocks(comm, rows, 1, starts, nrows, parts, gath, 2.0, -1.0, ys, rows, streams)
    torch.cuda.synchronize()
    assert stub_stats()["sends"] == g * (g - 1)
    ref = oracle.spmv(rows, rp, ci, v, xh, yh.copy(), 2.0, -1.0)
    for q in range(g):
        assert close(ys[q].cpu().numpy(), ref), q
    # --- fp32: the typed merges (RCCL_FLOAT32 = 7, receive offsets in bytes of the value type)
    M, n, g = 200, 12, 4
    rng = np.random.default_rng(9)
    starts, nrows = [0, 49, 100, 150], [50, 51, 50, 50]       # rows 49 and ... are shared by two blocks
    blocks_h = [rng.standard_normal(nrows[q] * n).astype(np.float32) for q in range(g)]
    C0 = rng.standard_normal(M * n).astype(np.float32)
    comm = sblas.comm_get([dev] * g)
    streams = [torch.cuda.Stream(device=td) for _ in range(g)]
    parts = [torch.from_numpy(b).to(td) for b in blocks_h]
    gath = [torch.empty(sum(nrows) * n, dtype=torch.float32, device=td) for _ in range(g)]
    Cs = [torch.from_numpy(C0.copy()).to(td) for _ in range(g)]
    torch.cuda.synchronize()
    stub_stats()
    sblas.merge_rowblocks_typed(comm, M, n, starts, nrows, parts, gath, 0.5, 2.0, Cs, M, streams)
    torch.cuda.synchronize()
    st = stub_stats()
    assert st["sends"] == g * (g - 1) and st["bytes"] == (g - 1) * sum(nrows) * n * 4, st
    want = 2.0 * C0.reshape(n, M).astype(np.float64)
    for q in range(g):
        want[:, starts[q]:starts[q] + nrows[q]] += 0.5 * blocks_h[q].reshape(n, nrows[q])
    for q in range(g):
        assert np.allclose(Cs[q].cpu().numpy().reshape(n, M), want, rtol=1e-5, atol=1e-5), q
    bufs = [torch.from_numpy(rng.standard_normal(1000).astype(np.float32)).to(td) for _ in range(g)]
    want = sum(b.cpu().numpy().astype(np.float64) for b in bufs)
    torch.cuda.synchronize()
    sblas.allreduce_sum_typed(comm, bufs, streams, 1000)
    torch.cuda.synchronize()
    assert stub_stats()["allreduces"] == 1
    for q in range(g):
        assert np.allclose(bufs[q].cpu().numpy(), want, rtol=1e-5, atol=1e-5), q
    print("COMM_STUB_OK")


if __name__ == "__main__":
    main()
