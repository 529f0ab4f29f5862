"""world_size = 2 rehearsal of the multi-GPU host logic on CPU (gloo): the product's placement functions
(sblas_partition_nnz / sblas_partition_dense through the C ABI) drive a two-rank method-2 merge (all-reduce of the
partial C, then C = beta*C + alpha*sum; and the packed row-block all-gather + scatter merge) and a method-1
column-block gather.  No GPU exists here, so each rank's
block product is computed by the oracle standing in for the device kernel -- what is under test is the placement,
the offsets/leading dimensions and the collective pattern bench.py and spmm.h use."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ASH85, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python"))
    import torch
    import torch.distributed as dist
    import oracle_py as O
    import sblas_amd as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, k, nnz, _, rp, ci, v = S.read_mtx(ASH85)
    N, alpha, beta = 64, 3.0, 4.0
    B = O.rand0to1(k * N)
    ref = np.ones(m * N)
    O.spmm(m, k, N, rp, ci, v, B, ref, alpha, beta)

    # ---- method 2: row block of A, full B, partial C summed over ranks ----
    part = S.partition_nnz(rp, world, rank)
    lo, cnt, r0 = part["first_nnz"], part["nnz"], part["start_row"]
    m_i = len(part["rowptr"]) - 1
    blk = np.zeros(m_i * N)
    O.spmm(m_i, k, N, part["rowptr"], ci[lo:lo + cnt].copy(), v[lo:lo + cnt].copy(), B, blk, 1.0, 1.0)
    ccopy = np.zeros((N, m))
    ccopy[:, r0:r0 + m_i] = blk.reshape(N, m_i)             # ldc = M, offset start_row (spmm.h:224-231)
    t = torch.from_numpy(ccopy.reshape(-1).copy())
    dist.all_reduce(t)                                       # spmm.h:260-262
    C2 = np.ones(m * N)
    O.lib().orc_axpby(C2.size, alpha, t.numpy(), beta, C2)   # kernel.h:27-38
    ok2 = bool(np.allclose(C2, ref, rtol=1e-10, atol=1e-12))

    # ---- method 2, fast merge (sblas_hip_merge_rowblocks_f64 / bench.py "method2_rowblocks"): packed
    # blocks (beta = 0, ldc = m_i), all-gather of equal-size padded buffers, scatter + alpha/beta with the partition
    # tables of every rank -- the boundary row of two neighbouring blocks receives two terms ----
    parts = [S.partition_nnz(rp, world, q) for q in range(world)]
    nrows_q = [len(p_["rowptr"]) - 1 for p_ in parts]
    maxblk = max(nrows_q) * N
    mine = np.zeros(maxblk)
    packed = np.zeros(m_i * N)
    O.spmm(m_i, k, N, part["rowptr"], ci[lo:lo + cnt].copy(), v[lo:lo + cnt].copy(), B, packed, 1.0, 0.0)
    mine[:m_i * N] = packed
    got = [torch.zeros(maxblk, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(got, torch.from_numpy(mine))
    acc = np.zeros((N, m))
    for q in range(world):
        blkq = got[q].numpy()[:nrows_q[q] * N].reshape(N, nrows_q[q])
        acc[:, parts[q]["start_row"]:parts[q]["start_row"] + nrows_q[q]] += blkq
    C3 = beta * np.ones(m * N) + alpha * acc.reshape(-1)
    ok3 = bool(np.allclose(C3, ref, rtol=1e-10, atol=1e-12))
    shared = [q for q in range(world - 1) if parts[q]["stop_row"] == parts[q + 1]["start_row"]]
    ok3 = ok3 and len(shared) > 0                            # ash85 at g = 2 cuts row 39: it is in both blocks

    # ---- method 1: column block of B and C, full A, gathered ----
    off, dim = S.partition_dense(N, world, rank)
    Cb = np.ones(m * dim)
    O.spmm(m, k, dim, rp, ci, v, B[off * k:(off + dim) * k].copy(), Cb, alpha, beta)
    pieces = [torch.zeros(m * S.partition_dense(N, world, r)[1], dtype=torch.float64) for r in range(world)]
    dist.all_gather(pieces, torch.from_numpy(Cb))
    C1 = torch.cat(pieces).numpy()
    ok1 = bool(np.allclose(C1, ref, rtol=1e-10, atol=1e-12))

    # ---- bench.py's timing reduction: max over ranks ----
    tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    okt = tt.item() == float(world)
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("%d %d %d %d" % (ok1, ok2, okt, ok3))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_two_rank_methods_over_gloo(sblas, oracle, tmp_path, world):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / ("rank%d.txt" % r)).read() == "1 1 1 1"
