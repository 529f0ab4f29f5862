#!/usr/bin/env python3
"""bench.py -- the S-BLAS CSR SpMM hot path on MI355X, measured as BASELINE.json asks.

Metric    : SpMM GFLOP/s (2*nnz*N / t) + achieved HBM GB/s, CSR x dense, N = 64 columns per GPU, fp64.
Workload  : BASELINE config 3 -- "nd24k" SpMM method 1, N = 64, alpha = beta = 1.  The SuiteSparse file is not
            in the image and cannot be fetched, so the default input is the synthetic stand-in of
            s-blas_amd/python/sblas_amd/synth.py (72 000 x 72 000, 399 nnz/row = 28 728 000 nnz, band +-2000,
            seed 211); pass --matrix file.mtx to use a real MatrixMarket file instead.  B is the reference's own
            (DenseMatrix ctor, matrix.h:519-528: srand(211), rand()/RAND_MAX in column-major order), C0 = 1.
A step    : one call of the drop-in boundary sblas_hip_spmm_csr_f64_i32 on device-resident inputs:
            stage 1 (B -> row-major staging copy + panel classifier) + stage 2 (row-panel SpMM, alpha/beta fused).
Multi-GPU : one process per GPU.  Method 1 partitions the dense columns and has no exchange step
            (spmm.h:83-161), so rank r multiplies the full A by its own 64-column block: per-GPU work is fixed,
            "scaling": "weak", no collective in the data path.
Extras    : (keys beside the contract's; the headline is never computed from them)
            cold_ms_per_step     the first ten steps after a pause (idle clocks), before the settling phase
            secondary            BASELINE config 5's shape on this GPU: Queen-like rows, 300 000 rows, N = 256
            product_merge        method 2 (spmm.h:163-284) through the PRODUCT's own merge (comm.hip: persistent RCCL
                                 communicator, packed row-block exchange / all-reduce), one process driving all GPUs like
                                 the reference: BASELINE config 4 (nd24k-like, N = 128) and config 5 (Queen-like, N = 256),
                                 kernel / merge / epilogue times separate, each with its own oracle check.  On one GPU
                                 the ranks are folded onto it (rehearsal of the code path, not a scaling figure).
            method2*             (N > 1) the same scheme with torch.distributed collectives, one process per GPU
            cpu_baseline         the oracle on this host's cores (N = 1 only)
Output    : exactly one JSON line on rank 0.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def source_sha16():
    """Fingerprint of the kernel sources: a committed PMC figure is only quoted for the code it was measured on."""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "s-blas_amd", "csrc", "*"))):
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(kernel, rows, nnz, n):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r*_hbm_traffic*.json; FETCH_SIZE and
    WRITE_SIZE in separate passes, gfx950 read correction applied there).  PMC counters cannot be read from inside
    this process, so the figure is only reported when the profile was taken on this very kernel source and workload."""
    sha = source_sha16()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic*.json")), reverse=True):
        try:
            d = json.load(open(path))
            w = d.get("workload", {})
            if (w.get("rows"), w.get("nnz"), w.get("n")) != (rows, nnz, n) or d.get("source_sha16") != sha:
                continue
            hits = [v for name, v in d["kernels"].items() if name == "sblas::" + kernel or name.startswith("sblas::" + kernel + "<")]
            if hits:
                return hits[0]["hbm_bytes_per_launch_corrected"], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def algorithmic_bytes(rows, cols, nnz, n, beta_nonzero=True):
    """nnz*(4+8) + (M+1)*4 + 8*K*N + (16 or 8)*M*N."""
    return nnz * 12 + (rows + 1) * 4 + 8 * cols * n + (16 if beta_nonzero else 8) * rows * n


def load_workload(args):
    import sblas_amd as S
    from sblas_amd import synth
    if args.matrix:
        rows, cols, nnz, _, rp, ci, v = S.read_mtx(args.matrix)
        name = os.path.basename(args.matrix)
    else:
        rows, (rp, ci, v) = synth.nd24k_like(scale=args.scale)
        cols, nnz = rows, int(rp[-1])
        name = "nd24k-like synthetic (M=K=%d, %d nnz/row, band +-2000, seed 211)" % (rows, 399)
    return name, rows, cols, nnz, rp, ci, v


def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    return O


def check_windows(O, got_view, rows, cols, n, rp, ci, v, Bh, total_steps, windows, alpha=1.0, c0=1.0):
    """C = c0 + total_steps * alpha * A*B on 64-row windows (first panel, a block boundary / the middle, the last rows)
    against the oracle; got_view: C as an (n, rows) tensor on the device."""
    worst = 0.0
    for r0 in windows:
        r0 = max(0, min(int(r0), rows - 64)) if rows >= 64 else 0
        r1 = min(rows, r0 + 64)
        ref = np.zeros(rows * n)
        O.spmm_rows(r0, r1, rows, cols, n, rp, ci, v, Bh, ref, alpha, 0.0)
        got = got_view[:, r0:r1].cpu().numpy()
        want = c0 + total_steps * ref.reshape(n, rows)[:, r0:r1]
        if not np.allclose(got, want, rtol=1e-9, atol=1e-9):
            return False, float(np.abs(got - want).max())
        worst = max(worst, float(np.abs(got - want).max()))
    return True, worst


def cpu_baseline(rows, cols, n, rp, ci, v, Bh, budget_s):
    """The oracle (CPU restatement of sblas_spmm_csr_cpu, spmm.h:56-68) on this host, one thread, same inputs.
    Runs whole passes over the workload until ~budget_s of CPU time has been spent (at least one)."""
    O = oracle()
    nnz = int(rp[-1])
    C = np.ones(rows * n)
    r_cal = max(1, rows // 16)
    t0 = time.perf_counter()
    O.spmm_rows(0, r_cal, rows, cols, n, rp, ci, v, Bh, C, 1.0, 1.0)
    t_cal = time.perf_counter() - t0
    est_full = t_cal * rows / r_cal
    passes = int(max(1, min(8, budget_s // max(est_full, 1e-9))))
    if est_full > 2 * budget_s:                     # huge matrix: a row prefix instead of full passes
        r_end = max(r_cal, int(rows * budget_s / est_full))
        C = np.ones(rows * n)
        t0 = time.perf_counter()
        O.spmm_rows(0, r_end, rows, cols, n, rp, ci, v, Bh, C, 1.0, 1.0)
        dt = time.perf_counter() - t0
        flops = 2.0 * float(rp[r_end]) * n
        sample = "rows [0,%d) of %d, one pass" % (r_end, rows)
    else:
        C = np.ones(rows * n)
        t0 = time.perf_counter()
        for _ in range(passes):
            O.spmm(rows, cols, n, rp, ci, v, Bh, C, 1.0, 1.0)
        dt = time.perf_counter() - t0
        flops = 2.0 * nnz * n * passes
        sample = "%d full pass(es) of the same workload (all %d rows)" % (passes, rows)
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    out = {"value": round(flops / dt / 1e9, 4), "unit": "GFLOP/s", "cores": 1, "kind": "port",
           "sample": sample, "seconds": round(dt, 2), "host_cpu": model, "host_threads_available": os.cpu_count()}
    try:                                            # second figure: the same loop with OpenMP over rows
        ncores = len(os.sched_getaffinity(0))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                ncores = max(1, min(ncores, int(int(quota) / int(period))))
        except Exception:
            pass
        os.environ.setdefault("OMP_NUM_THREADS", str(ncores))
        C = np.ones(rows * n)
        O.spmm_omp(rows, cols, n, rp, ci, v, Bh, C, 1.0, 1.0)
        reps, t0 = 0, time.perf_counter()
        while reps < 3 or (time.perf_counter() - t0 < 2.0 and reps < 50):
            O.spmm_omp(rows, cols, n, rp, ci, v, Bh, C, 1.0, 1.0)
            reps += 1
        dt2 = time.perf_counter() - t0
        out["all_cores"] = {"value": round(2.0 * nnz * n * reps / dt2 / 1e9, 3), "unit": "GFLOP/s",
                            "cores": int(os.environ["OMP_NUM_THREADS"]), "sample": "%d full passes, OpenMP over rows" % reps}
    except Exception as e:
        out["all_cores"] = {"error": str(e)}
    return out


def settle(torch, step, max_blocks=25, block=20):
    """Untimed steps until the device has left its idle clocks: after a pause an MI355X needs ~150 steps (50 ms) of
    this load before the step time stops falling (tools/graph_step.py).  Blocks of `block` steps, stop when two blocks
    in a row are no more than 1 % faster than the best before them; returns the steps run."""
    best, flat, done = None, 0, 0
    for _ in range(max_blocks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(block):
            step()
        e1.record()
        torch.cuda.synchronize()
        done += block
        ms = e0.elapsed_time(e1)
        if best is not None and ms > 0.99 * best:
            flat += 1
            if flat == 2:
                break
        else:
            flat = 0
        best = ms if best is None else min(best, ms)
    return done


def dominant_kernel_name(S, n, census):
    """Name (as rocprofv3 prints it, without the namespace) of the stage-2 kernel that did most of the panels."""
    ldbt = int(S.lib().sblas_hip_spmm_ldbt(n))
    if census["mfma"] >= max(census["windowed"], census["direct"]) and census["mfma"] > 0:
        return "spmm_mfma_kernel"
    if census["windowed"] >= census["direct"]:
        return "spmm_window6_kernel" if ldbt >= 64 else "spmm_lanes_kernel"
    if ldbt == 8:
        return "spmm (n=%d)" % n
    return "spmm_direct_dpp_kernel" if n > 32 else "spmm_direct_dpp_kernel<4>"


def bench_spmv(args, torch, S, dev, dist, world, rank, name, rows, cols, nnz, rp, ci, v):
    """y = A*x + y on every rank (replicas: SpMV has no column dimension to split), HIP-event kernel time."""
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rowptr, colidx, val = d(rp), d(ci), d(v)
    x = torch.ones(cols, dtype=torch.float64, device=dev)
    y = torch.ones(rows, dtype=torch.float64, device=dev)
    settled = 0 if args.no_settle else settle(torch, lambda: S.spmv(rows, cols, rowptr, colidx, val, x, 1.0, 1.0, y))
    for _ in range(args.warmup):
        S.spmv(rows, cols, rowptr, colidx, val, x, 1.0, 1.0, y)
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        S.spmv(rows, cols, rowptr, colidx, val, x, 1.0, 1.0, y)
        ev[k][1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.dist_backend == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    t_k = float(np.mean([e[0].elapsed_time(e[1]) for e in ev])) * 1e-3
    alg = nnz * 12 + (rows + 1) * 4 + 8 * cols + 16 * rows
    if rank == 0:
        rs = np.add.reduceat(v, rp[:-1].astype(np.int64)) if nnz else np.zeros(rows)
        want = 1.0 + (settled + args.warmup + args.steps) * rs
        if not np.allclose(y.cpu().numpy(), want, rtol=1e-9, atol=1e-9):
            raise SystemExit("spmv bench result mismatch")
        avg = nnz / max(rows, 1)
        sp_kernel = ("spmv_csr_lds_kernel" if avg > 96 else "spmv_csr_seg_kernel" if avg > 48 else
                     "spmv_csr_stream_kernel" if avg > 2.5 else "spmv_csr_kernel")
        sp_traffic, sp_src = (measured_traffic(sp_kernel, rows, nnz, 1)
                              if os.environ.get("SBLAS_SPMV_VARIANT", "") in ("", "auto") else (None, None))
        out = {"metric": "SpMV GFLOP/s (2*nnz/t), CSR fp64", "value": round(world * 2.0 * nnz * args.steps / elapsed / 1e9, 2),
               "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "settle_steps": settled,
               "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not args.matrix else "file",
               "config": {"workload": "SpMV y=A*x+y, %s, nnz=%d, replicas only" % (name, nnz), "rows": rows, "nnz": nnz},
               "roofline": {"bound": "hbm", "kernel": sp_kernel,
                            "achieved": round(alg / t_k / 1e9, 1),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / t_k / 1e9 / HBM_PEAK_GBS, 4),
                            "traffic": sp_traffic, "traffic_source": sp_src, "algorithmic_bytes_per_launch": alg,
                            "kernel_ms": round(t_k * 1e3, 5)},
               "cpu_baseline": None}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------
# secondary shape: BASELINE config 5's rows on one GPU
# ---------------------------------------------------------------------------------------------------------------
def secondary_queen(args, torch, S, dev, rows=300000, n=256, steps=10):
    """Queen_4147-like rows (SURVEY 8d stand-in: 20-30 clusters of 3 columns at 40 offsets scattered over +-50 000),
    300 000 rows, N = 256: every panel takes the direct kernel on 128-column tiles.  Own timing, roofline and oracle
    check; `grid` repeats it on the grid-structured variant (synth.queen_like_grid: the locality a 3-D FEM numbering
    has), where the classifier may hand panels to the matrix-core kernel."""
    from sblas_amd import synth
    O = oracle()
    out = {}
    for key, gen in (("scattered", lambda: synth.queen_like(rows)), ("grid", lambda: synth.queen_like_grid(rows))):
        rp, ci, v = gen()
        m = len(rp) - 1
        nnz = len(ci)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        rowptr, colidx, val = d(rp), d(ci), d(v)
        Bh = S.rand0to1(m * n)
        B = d(Bh)
        C = torch.ones(m * n, dtype=torch.float64, device=dev)
        ws = torch.empty(S.spmm_workspace_bytes(m, m, nnz, n) // 8, dtype=torch.float64, device=dev)
        step = lambda: S.spmm(m, m, rowptr, colidx, val, B, m, n, 1.0, 1.0, C, m, ws)
        S.panel_census()
        for _ in range(3):
            step()
        census = S.panel_census()
        census = {k: c // 3 for k, c in census.items()}
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        # the same step through a per-matrix plan (sblas_hip_spmm_plan_*: classified once, only the kernels with panels)
        plan = S.SpmmPlan(m, m, rowptr, colidx, n)
        pstep = lambda: plan.spmm(val, B, m, n, 1.0, 1.0, C, m, ws)
        for _ in range(3):
            pstep()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            pstep()
        e1.record()
        torch.cuda.synchronize()
        pms = e0.elapsed_time(e1) / steps
        pinfo = plan.info()
        plan.destroy()
        ok, err = check_windows(O, C.view(n, m), m, m, n, rp, ci, v, Bh, 2 * (3 + steps), (0, m // 2, m - 64))
        alg = algorithmic_bytes(m, m, nnz, n, True)
        kernel = dominant_kernel_name(S, n, census)
        traffic, src = measured_traffic(kernel, m, nnz, n)
        out[key] = {"workload": "Queen_4147-like (%s), %d rows, %d nnz, N=%d, alpha=beta=1" % (key, m, nnz, n),
                    "ms_per_step": round(ms, 5), "gflops": round(2.0 * nnz * n / ms / 1e6, 1), "panels": census,
                    "planned_ms_per_step": round(pms, 5), "plan": pinfo,
                    "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4), "traffic": traffic,
                                 "traffic_source": src, "algorithmic_bytes_per_launch": alg,
                                 "note": "whole step (staging + stage 2, two 128-column tiles), HIP events"},
                    "oracle_check": ok, "oracle_max_abs_diff": err}
        if key == "grid":
            # one rank's share of method 2 at g = 8 (row block 3, split by nonzeros): the call stages only the block's
            # column range of B, and the row-merging kernel follows the block's row-group phase
            a, b = (int(x) for x in np.searchsorted(rp, [nnz * 3 // 8, nnz * 4 // 8]))
            mi = b - a
            sub_rp = (rp[a:b + 1] - rp[a]).astype(np.int32)
            sub_ci, sub_v = ci[rp[a]:rp[b]], v[rp[a]:rp[b]]
            srp = d(sub_rp)
            Cb = torch.ones(mi * n, dtype=torch.float64, device=dev)
            bstep = lambda: S.spmm(mi, m, srp, colidx[rp[a]:rp[b]], val[rp[a]:rp[b]], B, m, n, 1.0, 1.0, Cb, mi, ws)
            for _ in range(3):
                bstep()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(steps):
                bstep()
            e1.record()
            torch.cuda.synchronize()
            bms = e0.elapsed_time(e1) / steps
            bci, bval = colidx[rp[a]:rp[b]].contiguous(), val[rp[a]:rp[b]].contiguous()
            bplan = S.SpmmPlan(mi, m, srp, bci, n)
            bpstep = lambda: bplan.spmm(bval, B, m, n, 1.0, 1.0, Cb, mi, ws)
            for _ in range(3):
                bpstep()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(steps):
                bpstep()
            e1.record()
            torch.cuda.synchronize()
            bpms = e0.elapsed_time(e1) / steps
            bpinfo = bplan.info()
            bplan.destroy()
            bok, berr = check_windows(O, Cb.view(n, mi), mi, m, n, sub_rp, sub_ci, sub_v, Bh, 2 * (3 + steps), (0, mi // 2, mi - 64))
            out[key]["method2_rank_share"] = {"block": "3 of 8 by nonzeros: rows [%d, %d)" % (a, b), "nnz": int(len(sub_ci)),
                                              "ms_per_step": round(bms, 5), "planned_ms_per_step": round(bpms, 5), "plan": bpinfo,
                                              "gflops": round(2.0 * len(sub_ci) * n / bms / 1e6, 1),
                                              "oracle_check": bok, "oracle_max_abs_diff": berr}
            del srp, Cb
        del rowptr, colidx, val, B, C, ws
        torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------------
# method 2 through the product's own merge: one process drives all GPUs (the reference's process model)
# ---------------------------------------------------------------------------------------------------------------
def product_method2(torch, S, O, devs, name, rows, cols, rp, ci, v, n, merge, steps, warmup):
    """sblas_spmm_csr_v2's device work (s-blas_amd/include/spmm.h) on g = len(devs) ranks from this one process:
    nnz row-block partition, per-rank SpMM on its own stream, then comm.hip's merge --
      rowblocks: packed m_i x N partial (beta = 0) + sblas_hip_merge_rowblocks_f64 (RCCL send/recv + scatter/alpha/beta)
      allreduce: zeroed M x N partial at row offset, ld = M (spmm.h:222-251) + sblas_hip_allreduce_sum_f64 (spmm.h:260-262)
                 + sblas_hip_axpby_f64 (spmm.h:283).
    devs all equal: ranks folded onto one device (rehearsal); distinct: RCCL over xGMI."""
    g, M, K = len(devs), rows, cols
    nnz = int(rp[-1])
    tdev = [torch.device("cuda", d) for d in devs]
    folded = len(set(devs)) == 1 and g > 1
    Bh = S.rand0to1(K * n)
    comm = S.comm_get(devs)
    parts = [S.partition_nnz(rp, g, q) for q in range(g)]
    starts = [p["start_row"] for p in parts]
    nrows = [len(p["rowptr"]) - 1 for p in parts]
    total_blocks = sum(nrows) * n
    A, Bs, Cs, streams, mstreams, part, gath, ws = [], [], [], [], [], [], [], []
    shared_B = {}
    piped = merge == "rowblocks_pipelined"
    T = 128
    ntiles = max(n // T, 1)
    tiles = [(c * T, (n - c * T) if c == ntiles - 1 else T) for c in range(ntiles)]   # the last tile takes the remainder
    for q in range(g):
        td = tdev[q]
        with torch.cuda.device(td):
            lo, k = parts[q]["first_nnz"], parts[q]["nnz"]
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(td)
            A.append((up(parts[q]["rowptr"]), up(ci[lo:lo + k]), up(v[lo:lo + k])))
            if devs[q] not in shared_B:
                shared_B[devs[q]] = up(Bh)
            Bs.append(shared_B[devs[q]])
            Cs.append(torch.ones(M * n, dtype=torch.float64, device=td))
            streams.append(torch.cuda.Stream(device=td))
            mstreams.append(torch.cuda.Stream(device=td))
            ws.append(torch.empty(max(S.spmm_workspace_bytes(nrows[q], K, k, n) // 8, 2), dtype=torch.float64, device=td))
            if merge == "allreduce":
                part.append(torch.zeros(M * n, dtype=torch.float64, device=td))
            else:   # rowblocks, rowblocks_pipelined
                part.append(torch.empty(max(nrows[q] * n, 1), dtype=torch.float64, device=td))
                gath.append(None if folded or g == 1 else torch.empty(max(total_blocks, 1), dtype=torch.float64, device=td))
    for d_ in set(devs):
        torch.cuda.synchronize(d_)
    ev = [[[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(g)] for _ in range(steps)]

    tl = [[torch.cuda.Event(enable_timing=True) for _ in range(1 + 2 * ntiles)] for _ in range(steps)]   # rank 0's timeline

    def step_piped(k=None):
        """spmm.h's column-tile pipeline: SpMM of tile c + 1 on the compute stream beside the exchange + scatter of tile c
        on the rank's second stream."""
        for q in range(g):
            if k is not None:
                ev[k][q][0].record(streams[q])
        if k is not None:
            tl[k][0].record(streams[0])
        for c, (c0, tc) in enumerate(tiles):
            for q in range(g):
                with torch.cuda.device(tdev[q]):
                    if nrows[q] > 0:
                        S.spmm(nrows[q], K, A[q][0], A[q][1], A[q][2], Bs[q][c0 * K:(c0 + tc) * K], K, tc, 1.0, 0.0,
                               part[q][c0 * nrows[q]:(c0 + tc) * nrows[q]], nrows[q], ws[q], stream=streams[q])
                    done = torch.cuda.Event()
                    done.record(streams[q])
                    mstreams[q].wait_event(done)
                    if k is not None and q == 0:
                        tl[k][1 + 2 * c].record(streams[0])
                    if k is not None and c == ntiles - 1:
                        ev[k][q][1].record(streams[q])
            gt = None if folded or g == 1 else [gath[q][c0 * sum(nrows):(c0 + tc) * sum(nrows)] for q in range(g)]
            S.merge_rowblocks(comm, M, tc, starts, nrows, [part[q][c0 * nrows[q]:(c0 + tc) * nrows[q]] for q in range(g)], gt,
                              1.0, 1.0, [Cs[q][c0 * M:(c0 + tc) * M] for q in range(g)], M, mstreams)
            if k is not None:
                tl[k][2 + 2 * c].record(mstreams[0])
        for q in range(g):
            with torch.cuda.device(tdev[q]):
                fin = torch.cuda.Event()
                fin.record(mstreams[q])
                streams[q].wait_event(fin)
                if k is not None:
                    ev[k][q][2].record(streams[q])
                    ev[k][q][3].record(streams[q])

    def step(k=None):
        if piped:
            return step_piped(k)
        for q in range(g):
            with torch.cuda.device(tdev[q]):
                st = streams[q]
                if merge == "allreduce":
                    with torch.cuda.stream(st):
                        part[q].zero_()
                if k is not None:
                    ev[k][q][0].record(st)
                if nrows[q] > 0:
                    if merge == "allreduce":
                        S.spmm(nrows[q], K, A[q][0], A[q][1], A[q][2], Bs[q], K, n, 1.0, 1.0, part[q], M, ws[q], stream=st,
                               c_offset=starts[q])
                    else:
                        S.spmm(nrows[q], K, A[q][0], A[q][1], A[q][2], Bs[q], K, n, 1.0, 0.0, part[q], nrows[q], ws[q], stream=st)
                if k is not None:
                    ev[k][q][1].record(st)
        if merge == "allreduce":
            S.allreduce_sum(comm, part, streams, M * n)
            for q in range(g):
                with torch.cuda.device(tdev[q]):
                    if k is not None:
                        ev[k][q][2].record(streams[q])
                    S.axpby(M * n, 1.0, part[q], 1.0, Cs[q], stream=streams[q])
                    if k is not None:
                        ev[k][q][3].record(streams[q])
        else:
            S.merge_rowblocks(comm, M, n, starts, nrows, part, gath if not (folded or g == 1) else None, 1.0, 1.0, Cs, M, streams)
            for q in range(g):
                with torch.cuda.device(tdev[q]):
                    if k is not None:
                        ev[k][q][2].record(streams[q])
                        ev[k][q][3].record(streams[q])

    def sync_all():
        for d_ in set(devs):
            torch.cuda.synchronize(d_)

    for _ in range(warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    sync_all()
    el = time.perf_counter() - t0
    ok, err = True, 0.0
    for q in sorted(set((0, g - 1))):
        cut = starts[min(1, g - 1)]
        okq, errq = check_windows(O, Cs[q].view(n, M), M, K, n, rp, ci, v, Bh, warmup + steps, (0, cut - 32, M // 2, M - 64))
        ok, err = ok and okq, max(err, errq)
    ms = lambda a, b: float(np.mean([[e[q][a].elapsed_time(e[q][b]) for q in range(g)] for e in ev], axis=0).max())
    out = {"workload": "%s, nnz=%d, N=%d, g=%d nnz row blocks, alpha=beta=1" % (name, nnz, n, g), "merge": merge,
           "devices": devs, "folded_onto_one_device": folded, "ms_per_step": round(el / steps * 1e3, 5),
           "gflops": round(2.0 * nnz * n * steps / el / 1e9, 1),
           "ms_spmm_max_over_ranks": round(ms(0, 1), 5),
           "ms_merge_max_over_ranks": round(ms(1, 2), 5),
           "ms_epilogue_max_over_ranks": round(ms(2, 3), 5),
           "merge_payload_bytes_per_rank": (M * n * 8) if merge == "allreduce" else (total_blocks * 8),
           "oracle_check": ok, "oracle_max_abs_diff": err,
           "api": ("sblas_hip_comm_get + sblas_hip_allreduce_sum_f64 + sblas_hip_axpby_f64" if merge == "allreduce" else
                   "sblas_hip_comm_get + sblas_hip_merge_rowblocks_f64 (the scatter / alpha / beta pass is part of the merge)")}
    if piped:
        rel = lambda j: round(float(np.mean([e[0].elapsed_time(e[j]) for e in tl])), 5)
        out["timeline_rank0_ms"] = [{"tile": c, "cols": tiles[c][1], "spmm_done": rel(1 + 2 * c), "merge_done": rel(2 + 2 * c)}
                                    for c in range(ntiles)]
        out["note"] = ("column-tile pipeline of sblas_spmm_csr_v2 (two streams per rank, events between them); ms_merge is the "
                       "tail behind the last SpMM.  " + ("Ranks folded onto one GPU: the streams share it, what the overlap is "
                       "worth over xGMI is unmeasured." if folded else ""))
    return out


def product_merge_sections(args, torch, S, devs4, devs8, rows, cols, rp, ci, v):
    """BASELINE config 4 (nd24k-like, N = 128) and config 5 (Queen-like, N = 256) through comm.hip, both merges."""
    from sblas_amd import synth
    O = oracle()
    res = {}
    for merge in ("rowblocks", "allreduce"):
        res["config4_" + merge] = product_method2(torch, S, O, devs4, "nd24k-like", rows, cols, rp, ci, v, 128, merge,
                                                  args.merge_steps, 2)
    qrows = args.queen_rows
    qrp, qci, qv = synth.queen_like(qrows)
    for merge in ("rowblocks", "rowblocks_pipelined", "allreduce"):
        res["config5_" + merge] = product_method2(torch, S, O, devs8, "Queen_4147-like (scattered), %d rows" % qrows, qrows, qrows,
                                                  qrp, qci, qv, 256, merge, max(2, args.merge_steps // 2), 1)
        torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-settle", action="store_true",
                    help="skip the untimed clock-settling steps in front of the warm-up (cold-start figure)")
    ap.add_argument("--ncols", type=int, default=64, help="dense columns per GPU (method 1)")
    ap.add_argument("--matrix", type=str, default=None, help="MatrixMarket file instead of the synthetic stand-in")
    ap.add_argument("--scale", type=float, default=1.0, help="row-count scale of the synthetic stand-in (rehearsal only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-method2", action="store_true", help="skip the torch.distributed method-2 sections (N > 1)")
    ap.add_argument("--no-extras", action="store_true", help="skip `secondary` and `product_merge` (profiling runs)")
    ap.add_argument("--merge-steps", type=int, default=6, help="timed steps of each product_merge section")
    ap.add_argument("--queen-rows", type=int, default=300000, help="rows of the Queen-like shape in the extras")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real thing); gloo + --fold-ranks rehearses the N>1 code path on one GPU")
    ap.add_argument("--fold-ranks", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--product-merge-child", type=str, default=None,
                    help="internal: run the product_merge sections on the given comma-separated devices and print their JSON")
    ap.add_argument("--extras-timeout", type=float, default=420.0, help="seconds the product_merge child process may take")
    ap.add_argument("--op", choices=["spmm", "spmv"], default="spmm",
                    help="spmm (the headline metric) or spmv (same matrix, x = y0 = 1; secondary measurement)")
    args = ap.parse_args()

    import torch
    import sblas_amd as S
    S.lib()   # fail loudly if the HIP library is missing

    if args.product_merge_child is not None:
        # child process of rank 0: the product's own merge over the job's GPUs, isolated so that a stuck collective on
        # hardware this code has never run on (comm.hip's distinct-device branches) cannot cost the headline line
        devs = [int(x) for x in args.product_merge_child.split(",")]
        torch.cuda.set_device(devs[0])
        name, rows, cols, nnz, rp, ci, v = load_workload(args)
        multi = len(set(devs)) > 1
        res = product_merge_sections(args, torch, S, devs if multi else devs[:1] * 4, devs if multi else devs[:1] * 8,
                                     rows, cols, rp, ci, v)
        print("PRODUCT_MERGE_JSON " + json.dumps(res), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if args.fold_ranks:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK=%d but only %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            dist.barrier()

    def all_ranks_ok(flag):
        """True only when `flag` holds on every rank (a rank-local failure must not strand the others in a collective)."""
        if dist is None:
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    name, rows, cols, nnz, rp, ci, v = load_workload(args)
    if args.op == "spmv":
        return bench_spmv(args, torch, S, dev, dist, world, rank, name, rows, cols, nnz, rp, ci, v)
    n = args.ncols
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rowptr, colidx, val = d(rp), d(ci), d(v)
    # the reference's B: K x (n * world) column-major, srand(211) / rand() in storage order (matrix.h:519-528); method 1
    # hands rank r the column block [r*n, (r+1)*n) (matrix.h:554-568)
    Bfull = S.rand0to1(cols * n * world)
    Bh = np.ascontiguousarray(Bfull[rank * cols * n:(rank + 1) * cols * n])
    del Bfull
    B = d(Bh)
    C = torch.ones(rows * n, dtype=torch.float64, device=dev)
    Bt = torch.empty(S.spmm_workspace_bytes(rows, cols, nnz, n) // 8, dtype=torch.float64, device=dev)   # the C ABI's workspace

    def step():
        # the drop-in boundary itself: sblas_hip_spmm_csr_f64_i32 (staging + classifier in one launch, then stage 2)
        S.spmm(rows, cols, rowptr, colidx, val, B, cols, n, 1.0, 1.0, C, rows, Bt)

    # cold figure: the first ten steps this process issues (idle clocks; what a caller making a handful of calls sees)
    torch.cuda.synchronize()
    time.sleep(0.5)
    S.panel_census()
    c0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - c0) / 10 * 1e3
    census = {k: c // 10 for k, c in S.panel_census().items()}
    cold_steps = 10

    settled = 0 if args.no_settle else settle(torch, step)
    if dist is not None and not args.no_settle:
        # ranks settle after different step counts: line them up, then 40 more steps each, so that no rank sits idle
        # (and drops its clocks again) for tens of milliseconds in front of the timed region
        barrier()
        for _ in range(40):
            step()
        settled += 40
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel times, ten extra steps straight after the timed region (clocks still settled) through the SPLIT entry
    # points (staging | classifier + stage 2 -- the same kernels, one launch more than the fused entry): an event
    # between the stages, and the launcher's own HIP events around the dominant stage-2 launch alone (a diagnostic
    # hook of the C ABI).  Through the fused entry the hook's first event would sit straight behind the staging
    # launch, and the L2 write-back of the 37 MB it wrote (~10 us) would be billed to the kernel behind it.
    t_dom, extra_steps, samples = None, 0, []
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(10)]
    try:
        S.kernel_events(True)
        for k in range(10):
            ev[k][0].record()
            S.dense_to_rowmajor(cols, n, B, cols, Bt)
            ev[k][1].record()
            S.spmm_rowmajorB(rows, cols, rowptr, colidx, val, Bt, n, 1.0, 1.0, C, rows)
            ev[k][2].record()
            extra_steps += 1
            try:
                samples.append(S.last_kernel_ms())
            except S.SblasError:
                pass                                # (n <= 32 column tiers never launch the instrumented kernel)
    finally:
        S.kernel_events(False)
    torch.cuda.synchronize()
    t_dom = float(np.mean(samples)) * 1e-3 if samples else None
    t_stage1 = float(np.mean([e[0].elapsed_time(e[1]) for e in ev])) * 1e-3
    s2 = [e[1].elapsed_time(e[2]) for e in ev]
    t_stage2 = float(np.mean(s2)) * 1e-3
    # the same step through a per-matrix plan (informational: `value` is the unplanned call, the drop-in boundary as a
    # caller that knows nothing of plans uses it)
    planned = None
    try:
        t_c0 = time.perf_counter()
        plan = S.SpmmPlan(rows, cols, rowptr, colidx, n)
        torch.cuda.synchronize()
        create_ms = (time.perf_counter() - t_c0) * 1e3
        for _ in range(10):
            plan.spmm(val, B, cols, n, 1.0, 1.0, C, rows, Bt)
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        for _ in range(args.steps):
            plan.spmm(val, B, cols, n, 1.0, 1.0, C, rows, Bt)
        torch.cuda.synchronize()
        planned = {"ms_per_step": round((time.perf_counter() - p0) / args.steps * 1e3, 5), "plan_create_ms": round(create_ms, 3),
                   "plan": plan.info(), "note": "sblas_hip_spmm_csr_f64_i32_planned: panels classified once, only the kernels with "
                                                "panels launched; same results bit for bit"}
        extra_steps += 10 + args.steps
        plan.destroy()
    except S.SblasError as ex:
        planned = {"error": repr(ex)}
    kernel = dominant_kernel_name(S, n, census)
    if not (kernel.startswith("spmm_window6") or kernel.startswith("spmm_lanes")):
        t_dom = None                                 # the launcher's events bracket the LDS-tiled kernels only
    t_roof = t_dom if t_dom else t_stage2

    # correctness guard on this rank's result: C = 1 + steps_total * A*B on 64-row windows (first panel, a middle one,
    # the last rows) vs the oracle
    total_steps = cold_steps + settled + args.warmup + args.steps + extra_steps
    check, check_err = None, None
    failures = []                                   # result mismatches found on rank 0 (fatal, reported at the end)
    if rank == 0:
        O = oracle()
        check, check_err = check_windows(O, C.view(n, rows), rows, cols, n, rp, ci, v, Bh, total_steps, (0, rows // 3, rows - 64))
        if not check:
            failures.append("bench result does not match the oracle: max diff %g" % check_err)
            if world == 1:
                raise SystemExit(failures[0])

    flops_step = 2.0 * nnz * n                      # per GPU
    value = world * flops_step * args.steps / elapsed / 1e9
    alg = algorithmic_bytes(rows, cols, nnz, n, True)
    traffic, traffic_src = (measured_traffic(kernel, rows, nnz, n)
                            if os.environ.get("SBLAS_SPMM_VARIANT", "") in ("", "auto") else (None, None))
    out = {
        "metric": "SpMM GFLOP/s (2*nnz*N/t), CSR x dense N=64, fp64",
        "value": round(value, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "settle_steps": settled,   # untimed steps in front of the warm-up until the clocks have settled (see settle())
        "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not args.matrix else "file",
        "config": {"workload": "SpMM method-1 (dense-B column partition), %s, nnz=%d, N=%d columns per GPU, alpha=beta=1, "
                               "B = srand(211)/rand() as the reference's DenseMatrix, inputs resident in HBM; "
                               "step = B->row-major staging + row-panel SpMM" % (name, nnz, n),
                   "rows": rows, "cols": cols, "nnz": nnz, "n_cols_per_gpu": n, "parallelism": "method1-colblock x%d" % world},
        "roofline": {"bound": "hbm", "kernel": kernel,
                     "achieved": round(alg / t_roof / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / t_roof / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": traffic_src, "source_sha16": source_sha16(),
                     "algorithmic_bytes_per_launch": alg, "kernel_ms": round(t_roof * 1e3, 5),
                     "kernel_ms_source": "HIP events around the one launch, 10 steps" if t_dom else "HIP events around stage 2",
                     "stage2_ms": round(t_stage2 * 1e3, 5), "stage2_ms_median": round(float(np.median(s2)), 5),
                     "stage2_ms_min": round(float(np.min(s2)), 5),
                     "staging_kernel_ms": round(t_stage1 * 1e3, 5),
                     "kernel_gflops": round(flops_step / t_roof / 1e9, 1), "panels": census},
        "hbm_gbs_whole_step": round(alg / (elapsed / args.steps) / 1e9, 1),
        "planned": planned,
        "cold_ms_per_step": round(cold_ms, 5),      # first ten steps of the process, idle clocks, host wall incl. launches
        "oracle_check": check, "oracle_max_abs_diff": check_err,
    }

    # ---- method 1, STRONG scaling: BASELINE's metric is N = 64 in total, and method 1 hands GPU i the columns
    # [i * ceil(64 / g), ...) of B and C with the full A (matrix.h:554-568): 32 / 16 / 8 columns per GPU at g = 2 / 4 / 8.
    # No collective.  With one GPU the same widths are timed one after the other on it (`method1_widths`): the per-GPU
    # step a g-GPU run sees, measured, not a projection of the g-GPU figure.
    def time_width(nw, col0, steps, settle_steps):
        Bref = S.rand0to1(cols * 64)
        Bw_h = np.ascontiguousarray(Bref[col0 * cols:(col0 + nw) * cols])
        del Bref
        Bw = d(Bw_h)
        Cw = torch.ones(rows * nw, dtype=torch.float64, device=dev)
        wsw = torch.empty(S.spmm_workspace_bytes(rows, cols, nnz, nw) // 8, dtype=torch.float64, device=dev)
        stepw = lambda: S.spmm(rows, cols, rowptr, colidx, val, Bw, cols, nw, 1.0, 1.0, Cw, rows, wsw)
        S.panel_census()
        stepw()
        torch.cuda.synchronize()
        cen = S.panel_census()
        for _ in range(settle_steps):
            stepw()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            stepw()
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t0
        ok, err = check_windows(oracle(), Cw.view(nw, rows), rows, cols, nw, rp, ci, v, Bw_h, 1 + settle_steps + steps,
                                (0, rows // 3, rows - 64))
        return el, ok, err, cen

    if n == 64 and world > 1:
        off, dim = S.partition_dense(64, world, rank)
        el_s, ok_s, err_s, cen_s = time_width(dim, off, args.steps, 60)
        t = torch.tensor([el_s, 0.0 if ok_s else 1.0], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el_s, bad_s = float(t[0].item()), bool(t[1].item() > 0.5)
        if bad_s:
            failures.append("method-1 strong-scaling result does not match the oracle on some rank (rank 0: %g)" % err_s)
        out["method1_strong"] = {
            "scaling": "strong", "n_total_cols": 64, "cols_per_gpu": dim, "oracle_check": not bad_s,
            "gflops": round(2.0 * nnz * 64 * args.steps / el_s / 1e9, 2), "ms_per_step": round(el_s / args.steps * 1e3, 5),
            "speedup_vs_this_runs_per_gpu_n64_step": round((elapsed / args.steps) / (el_s / args.steps), 3),
            "panels_rank0": cen_s,
            "note": "N = 64 total over %d GPUs, no collective; every GPU streams all of A (ideal speed-up at g = 8: 455.5 / 358.7 MB = 1.27x by HBM bytes, SURVEY 8d)" % world,
        }
    elif n == 64 and world == 1 and not args.no_extras:
        widths = {}
        for g_, nw in ((2, 32), (4, 16), (8, 8)):
            el_w, ok_w, err_w, cen_w = time_width(nw, 0, args.steps, 100)
            if not ok_w:
                failures.append("method-1 width %d does not match the oracle: max diff %g" % (nw, err_w))
            widths["g%d" % g_] = {"cols_per_gpu": nw, "ms_per_step": round(el_w / args.steps * 1e3, 5),
                                  "per_gpu_gflops": round(2.0 * nnz * nw * args.steps / el_w / 1e9, 2),
                                  "job_gflops_if_every_gpu_ran_this_step": round(2.0 * nnz * 64 * args.steps / el_w / 1e9, 2),
                                  "oracle_check": ok_w, "panels": cen_w}
        out["method1_widths"] = {"note": "one GPU, the column widths method 1 gives each of g GPUs at N = 64 (matrix.h:554-568); "
                                         "measured one after the other on this GPU", **widths}

    # ---- method 2 with torch.distributed collectives (one process per GPU), informational -------------------------------
    if world > 1 and not args.no_method2:
        m2_error = None
        try:
            part = S.partition_nnz(rp, world, rank)
            parts = [S.partition_nnz(rp, world, q) for q in range(world)]
        except Exception as ex:                      # rank-local: agree before any collective
            part, parts, m2_error = None, None, repr(ex)
        if all_ranks_ok(m2_error is None):
            lo, k = part["first_nnz"], part["nnz"]
            rp_i, ci_i, v_i = d(part["rowptr"]), colidx[lo:lo + k].contiguous(), val[lo:lo + k].contiguous()
            m_i = len(part["rowptr"]) - 1
            B2h = S.rand0to1(cols * n)                                 # replicated B
            B2 = d(B2h)
            C2 = torch.ones(rows * n, dtype=torch.float64, device=dev)
            Ccopy = torch.zeros(rows * n, dtype=torch.float64, device=dev)
            e = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]

            def step2(k=None):
                Ccopy.zero_()                                     # spmm.h:182-183 (zero buffer), on device
                if k is not None: e[k][0].record()
                S.dense_to_rowmajor(cols, n, B2, cols, Bt)
                if m_i > 0:
                    S.spmm_rowmajorB(m_i, cols, rp_i, ci_i, v_i, Bt, n, 1.0, 1.0, Ccopy, rows, c_offset=part["start_row"])
                if k is not None: e[k][1].record()
                if args.dist_backend == "nccl":
                    dist.all_reduce(Ccopy)                        # spmm.h:260-262, RCCL over xGMI
                else:                                             # rehearsal: gloo on a host copy
                    h = Ccopy.cpu()
                    dist.all_reduce(h)
                    Ccopy.copy_(h)
                if k is not None: e[k][2].record()
                S.axpby(rows * n, 1.0, Ccopy, 1.0, C2)            # spmm.h:283 -> kernel.h:27-38
                if k is not None: e[k][3].record()

            for _ in range(args.warmup):
                step2()
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            for k in range(args.steps):
                step2(k)
            torch.cuda.synchronize()
            barrier()
            el2 = time.perf_counter() - t0
            t = torch.tensor([el2], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
            m2_ok = None
            wins = (0, parts[min(1, world - 1)]["start_row"] - 32, rows // 2, rows - 64)
            if rank == 0:
                m2_ok, m2_err = check_windows(oracle(), C2.view(n, rows), rows, cols, n, rp, ci, v, B2h, args.warmup + args.steps, wins)
                if not m2_ok:
                    failures.append("method-2 bench result does not match the oracle: max diff %g" % m2_err)
            out["method2"] = {
                "oracle_check": m2_ok,
                "scaling": "strong", "n_total_cols": n, "gflops": round(flops_step * args.steps / el2 / 1e9, 2),
                "ms_per_step": round(el2 / args.steps * 1e3, 5),
                "ms_spmm": round(float(np.mean([x[0].elapsed_time(x[1]) for x in e])), 5),
                "ms_allreduce": round(float(np.mean([x[1].elapsed_time(x[2]) for x in e])), 5),
                "ms_axpby": round(float(np.mean([x[2].elapsed_time(x[3]) for x in e])), 5),
                "allreduce_payload_bytes": rows * n * 8,
                "note": "rank-0 stage times; merge = torch.distributed all_reduce (RCCL) on the full M x N buffer as spmm.h:260-262",
            }

            # packed row blocks, all-gather, one scatter + alpha/beta pass with torch.distributed
            starts = [p_["start_row"] for p_ in parts]
            nrows = [len(p_["rowptr"]) - 1 for p_ in parts]
            maxblk = max(max(nrows), 1) * n
            mine = torch.zeros(maxblk, dtype=torch.float64, device=dev)
            allb = torch.zeros(world * maxblk, dtype=torch.float64, device=dev)
            C3 = torch.ones(rows * n, dtype=torch.float64, device=dev)
            e3 = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]

            def step3(k=None):
                if k is not None: e3[k][0].record()
                S.dense_to_rowmajor(cols, n, B2, cols, Bt)
                if m_i > 0:
                    S.spmm_rowmajorB(m_i, cols, rp_i, ci_i, v_i, Bt, n, 1.0, 0.0, mine, m_i)   # beta = 0: no zero fill
                if k is not None: e3[k][1].record()
                if args.dist_backend == "nccl":
                    dist.all_gather_into_tensor(allb, mine)
                else:
                    hs = [torch.empty(maxblk, dtype=torch.float64) for _ in range(world)]
                    dist.all_gather(hs, mine.cpu())
                    allb.copy_(torch.cat(hs))
                if k is not None: e3[k][2].record()
                S.merge_rowblocks_local(rows, n, starts, nrows, [allb[q * maxblk:(q + 1) * maxblk] for q in range(world)],
                                        1.0, 1.0, C3)
                if k is not None: e3[k][3].record()

            for _ in range(args.warmup):
                step3()
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            for k in range(args.steps):
                step3(k)
            torch.cuda.synchronize()
            barrier()
            el3 = time.perf_counter() - t0
            t = torch.tensor([el3], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el3 = float(t.item())
            m3_ok = None
            if rank == 0:
                m3_ok, m3_err = check_windows(oracle(), C3.view(n, rows), rows, cols, n, rp, ci, v, B2h, args.warmup + args.steps, wins)
                if not m3_ok:
                    failures.append("method-2 (row-block merge) bench result does not match the oracle: max diff %g" % m3_err)
            out["method2_rowblocks"] = {
                "oracle_check": m3_ok, "scaling": "strong", "n_total_cols": n,
                "gflops": round(flops_step * args.steps / el3 / 1e9, 2), "ms_per_step": round(el3 / args.steps * 1e3, 5),
                "ms_spmm": round(float(np.mean([x[0].elapsed_time(x[1]) for x in e3])), 5),
                "ms_allgather": round(float(np.mean([x[1].elapsed_time(x[2]) for x in e3])), 5),
                "ms_merge": round(float(np.mean([x[2].elapsed_time(x[3]) for x in e3])), 5),
                "allgather_payload_bytes_per_rank": maxblk * 8,
                "note": "packed row blocks (beta = 0), torch.distributed all_gather_into_tensor (RCCL), sblas_hip_merge_rowblocks_local_f64",
            }
            del B2, C2, Ccopy, mine, allb, C3
        else:
            out["method2_error"] = m2_error or "another rank failed to partition"

    # ---- extras: rank 0 alone drives them (the other ranks wait on the store, CPU side, GPUs idle) -------------------------
    if not args.no_extras:
        store = None
        if dist is not None:
            store = dist.distributed_c10d._get_default_store()
            barrier()
        if rank == 0:
            del rowptr, colidx, val, B, C, Bt
            torch.cuda.empty_cache()
            try:
                if world == 1:
                    out["secondary"] = secondary_queen(args, torch, S, dev, rows=args.queen_rows)
                    for key, sec in out["secondary"].items():
                        if not sec["oracle_check"]:
                            failures.append("secondary (%s) does not match the oracle: max diff %g" % (key, sec["oracle_max_abs_diff"]))
                        share = sec.get("method2_rank_share")
                        if share and not share["oracle_check"]:
                            failures.append("secondary (%s) row block does not match the oracle: max diff %g" % (key, share["oracle_max_abs_diff"]))
                # the product's own merge: all GPUs of the job driven from ONE process (a child of rank 0, so that a
                # hang on untried hardware paths costs a timeout, not the line); one GPU: folded ranks (g = 4 / 8)
                ndev = 1 if args.fold_ranks else min(world, torch.cuda.device_count())
                devs = list(range(ndev)) if ndev > 1 else [local_rank]
                import subprocess
                cmd = [sys.executable, os.path.abspath(__file__), "--product-merge-child", ",".join(str(d_) for d_ in devs),
                       "--scale", str(args.scale), "--merge-steps", str(args.merge_steps), "--queen-rows", str(args.queen_rows)]
                if args.matrix:
                    cmd += ["--matrix", args.matrix]
                env = {k_: v_ for k_, v_ in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
                try:
                    cp = subprocess.run(cmd, capture_output=True, text=True, timeout=args.extras_timeout, env=env)
                    lines = [l for l in cp.stdout.splitlines() if l.startswith("PRODUCT_MERGE_JSON ")]
                    if cp.returncode == 0 and lines:
                        out["product_merge"] = json.loads(lines[-1][len("PRODUCT_MERGE_JSON "):])
                    else:
                        out["product_merge_error"] = "child rc=%d: %s" % (cp.returncode, (cp.stderr or cp.stdout)[-400:])
                except subprocess.TimeoutExpired:
                    out["product_merge_error"] = "child process exceeded %.0f s and was stopped" % args.extras_timeout
                for key, sec in out.get("product_merge", {}).items():
                    if not sec["oracle_check"]:
                        failures.append("product_merge %s does not match the oracle: max diff %g" % (key, sec["oracle_max_abs_diff"]))
            except Exception as ex:                  # extras never cost the headline line
                out["extras_error"] = repr(ex)
            if store is not None:
                store.set("sblas_extras_done", "1")
        elif store is not None:
            import datetime   # (explicit: the store's own default may be shorter than the child's limit)
            store.wait(["sblas_extras_done"], datetime.timedelta(seconds=args.extras_timeout + 180.0))

    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(rows, cols, n, rp, ci, v, Bh, args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None   # reported at N=1 only
    if rank == 0 and not failures:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if failures:                                    # a wrong result is never reported as a measurement
        raise SystemExit("; ".join(failures))


if __name__ == "__main__":
    main()
