#!/usr/bin/env python3
"""bench.py -- the S-BLAS CSR SpMM hot path on MI355X, measured as BASELINE.json asks.

Metric    : SpMM GFLOP/s (2*nnz*N / t) + achieved HBM GB/s, CSR x dense, N = 64 columns per GPU, fp64.
Workload  : BASELINE config 3 -- "nd24k" SpMM method 1, N = 64, alpha = beta = 1.  The SuiteSparse file is not
            in the image and cannot be fetched, so the default input is the synthetic stand-in of
            s-blas_amd/python/sblas_amd/synth.py (72 000 x 72 000, 399 nnz/row = 28 728 000 nnz, band +-2000,
            seed 211); pass --matrix file.mtx to use a real MatrixMarket file instead.
A step    : one full pass of the hot path through the C ABI on device-resident inputs:
            stage 1 (B -> row-major staging copy) + stage 2 (row-panel SpMM, alpha/beta fused).
Multi-GPU : one process per GPU.  Method 1 partitions the dense columns and has no exchange step
            (spmm.h:83-161), so rank r multiplies the full A by its own 64-column block: per-GPU work is fixed,
            "scaling": "weak", no collective in the data path.  With --gpus > 1 the same run ALSO times method 2
            (row-block A, RCCL all-reduce of the partial C, fused axpby; spmm.h:163-284) on N = 64 total columns
            and reports it under "method2" (strong scaling, merge time separate) -- informational.
Output    : exactly one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)

# stage-2 kernel launched for a 64-column block under each SBLAS_SPMM_VARIANT (s-blas_amd/csrc/kernels.hip)
KERNEL_OF_VARIANT = {"": "spmm_window6_kernel", "auto": "spmm_window6_kernel", "win6": "spmm_window6_kernel", "win5": "spmm_window5_kernel<false>", "win3": "spmm_window3_kernel<7>",
                     "win2": "spmm_window2_kernel<7>", "win4": "spmm_window4_kernel<false>",
                     "dpp": "spmm_direct_dpp_kernel<2>", "direct": "spmm_rowpanel_kernel",
                     "win32": "spmm_window_kernel<2,64,8>", "win64": "spmm_window_kernel<4,128,4>",
                     "win128": "spmm_window_kernel<8,128,4>", "win64w64": "spmm_window_kernel<4,64,8>",
                     "win32w128": "spmm_window_kernel<2,128,4>"}


def measured_traffic(kernel, rows, nnz, n):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r*_hbm_traffic.json; FETCH_SIZE and
    WRITE_SIZE in separate passes, gfx950 read correction applied there).  PMC counters cannot be read from inside
    this process, so the figure is only reported when the profile was taken on this very kernel and workload."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic*.json")), reverse=True):
        try:
            d = json.load(open(path))
            w = d.get("workload", {})
            if (w.get("rows"), w.get("nnz"), w.get("n")) != (rows, nnz, n):
                continue
            # (template arguments chosen inside the library, e.g. the groups per wave, are part of the profiled name)
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


def cpu_baseline(rows, cols, n, rp, ci, v, Bh, budget_s):
    """The oracle (CPU restatement of sblas_spmm_csr_cpu, spmm.h:56-68) on this host, one thread, same inputs.
    Runs whole passes over the workload until ~budget_s of CPU time has been spent (at least one)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    nnz = int(rp[-1])
    C = np.ones(rows * n)
    # calibrate on 1/16 of the rows, then pick the number of full passes
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
    # second figure: the same loop with rows spread over the cores this process may use
    try:
        ncores = len(os.sched_getaffinity(0))
        try:                                                         # a container's CPU quota, if tighter
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                ncores = max(1, min(ncores, int(int(quota) / int(period))))
        except Exception:
            pass
        os.environ.setdefault("OMP_NUM_THREADS", str(ncores))
        C = np.ones(rows * n)
        O.spmm_omp(rows, cols, n, rp, ci, v, Bh, C, 1.0, 1.0)          # warm-up (thread pool, page faults)
        reps, t0 = 0, time.perf_counter()
        while reps < 3 or (time.perf_counter() - t0 < 2.0 and reps < 50):
            O.spmm_omp(rows, cols, n, rp, ci, v, Bh, C, 1.0, 1.0)
            reps += 1
        dt2 = time.perf_counter() - t0
        out["all_cores"] = {"value": round(2.0 * nnz * n * reps / dt2 / 1e9, 3), "unit": "GFLOP/s",
                            "cores": int(os.environ["OMP_NUM_THREADS"]), "sample": "%d full passes, OpenMP over rows" % reps}
    except Exception as e:                                              # never let the extra figure break the run
        out["all_cores"] = {"error": str(e)}
    return out


def settle(torch, step, max_blocks=25, block=20):
    """Untimed steps until the device has left its idle clocks: after a pause an MI355X needs ~150 steps (50 ms) of
    this load before the step time stops falling (0.33 -> 0.27 ms, tools/graph_step.py).  Blocks of `block` steps,
    stop when two blocks in a row are no more than 1 % faster than the best before them; returns the steps run."""
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
    # y = 1 + steps_total * A*1  -> compare with row sums (exact structure check)
    if rank == 0:
        rs = np.add.reduceat(v, rp[:-1].astype(np.int64)) if nnz else np.zeros(rows)
        want = 1.0 + (settled + args.warmup + args.steps) * rs
        if not np.allclose(y.cpu().numpy(), want, rtol=1e-9, atol=1e-9):
            raise SystemExit("spmv bench result mismatch")
        sp_kernel = ("spmv_csr_lds_kernel" if nnz > 96 * rows and os.environ.get("SBLAS_SPMV_VARIANT", "") in ("", "auto", "lds")
                     else "spmv_csr_kernel")
        sp_traffic, sp_src = (measured_traffic(sp_kernel, rows, nnz, 1)
                              if os.environ.get("SBLAS_SPMV_VARIANT", "") in ("", "auto") else (None, None))
        out = {"metric": "SpMV GFLOP/s (2*nnz/t), CSR fp64", "value": round(world * 2.0 * nnz * args.steps / elapsed / 1e9, 2),
               "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "settle_steps": settled,
               "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not args.matrix else "file",
               "config": {"workload": "SpMV y=A*x+y, %s, nnz=%d, replicas only" % (name, nnz), "rows": rows, "nnz": nnz},
               "roofline": {"bound": "hbm",
                            "kernel": sp_kernel,
                            "achieved": round(alg / t_k / 1e9, 1),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / t_k / 1e9 / HBM_PEAK_GBS, 4),
                            "traffic": sp_traffic, "traffic_source": sp_src, "algorithmic_bytes_per_launch": alg,
                            "kernel_ms": round(t_k * 1e3, 5)},
               "cpu_baseline": None}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


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
    ap.add_argument("--no-method2", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real thing); gloo + --fold-ranks rehearses the N>1 code path on one GPU")
    ap.add_argument("--fold-ranks", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--op", choices=["spmm", "spmv"], default="spmm",
                    help="spmm (the headline metric) or spmv (same matrix, x = y0 = 1; secondary measurement)")
    args = ap.parse_args()

    import torch
    import sblas_amd as S
    S.lib()   # fail loudly if the HIP library is missing

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

    name, rows, cols, nnz, rp, ci, v = load_workload(args)
    if args.op == "spmv":
        return bench_spmv(args, torch, S, dev, dist, world, rank, name, rows, cols, nnz, rp, ci, v)
    n = args.ncols
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rowptr, colidx, val = d(rp), d(ci), d(v)
    gen = torch.Generator(device="cpu").manual_seed(211 + rank)
    Bh = torch.rand(cols * n, dtype=torch.float64, generator=gen)
    B = Bh.to(dev)
    C = torch.ones(rows * n, dtype=torch.float64, device=dev)
    ldbt = int(S.lib().sblas_hip_spmm_ldbt(n))
    Bt = torch.empty(S.spmm_workspace_bytes(rows, cols, nnz, n) // 8, dtype=torch.float64, device=dev)   # the C ABI's workspace

    def step():
        # the drop-in boundary itself: sblas_hip_spmm_csr_f64_i32 (staging + classifier in one launch, then stage 2)
        S.spmm(rows, cols, rowptr, colidx, val, B, cols, n, 1.0, 1.0, C, rows, Bt)

    settled = 0 if args.no_settle else settle(torch, step)
    if dist is not None and not args.no_settle:
        # ranks settle after different step counts (140-220): line them up, then 40 more steps each, so that no rank
        # sits idle (and drops its clocks again) for tens of milliseconds in front of the timed region
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
                pass                                # (variants that never launch a windowed kernel)
    finally:
        S.kernel_events(False)
    torch.cuda.synchronize()
    t_dom = float(np.mean(samples)) * 1e-3 if samples else None
    t_stage1 = float(np.mean([e[0].elapsed_time(e[1]) for e in ev])) * 1e-3
    s2 = [e[1].elapsed_time(e[2]) for e in ev]
    t_stage2 = float(np.mean(s2)) * 1e-3
    t_roof = t_dom if t_dom else t_stage2

    # correctness guard on this rank's result: C = 1 + (warmup+steps) * A*B on 64 sampled rows vs the oracle
    total_steps = settled + args.warmup + args.steps + extra_steps
    check = None
    failures = []                                   # result mismatches found on rank 0 (fatal, reported at the end)
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py as O
        r0 = rows // 3
        ref = np.zeros(rows * n)
        O.spmm_rows(r0, r0 + 64, rows, cols, n, rp, ci, v, Bh.numpy(), ref, 1.0, 0.0)
        got = C.view(n, rows)[:, r0:r0 + 64].cpu().numpy()
        want = 1.0 + total_steps * ref.reshape(n, rows)[:, r0:r0 + 64]
        check = bool(np.allclose(got, want, rtol=1e-9, atol=1e-9))
        if not check and not os.environ.get("SBLAS_ABLATE"):   # (SBLAS_ABLATE: diagnostic builds compute garbage on purpose)
            msg = "bench result does not match the oracle: max diff %g" % np.abs(got - want).max()
            if world == 1:
                raise SystemExit(msg)
            failures.append(msg)                    # N > 1: leaving now would strand the other ranks in a collective


    flops_step = 2.0 * nnz * n                      # per GPU
    value = world * flops_step * args.steps / elapsed / 1e9
    alg = algorithmic_bytes(rows, cols, nnz, n, True)
    kernel = KERNEL_OF_VARIANT.get(os.environ.get("SBLAS_SPMM_VARIANT", ""), "spmm_direct_dpp_kernel<2>") if n > 32 and n <= 64 else "spmm (n=%d)" % n
    # (the committed PMC passes were taken with the default kernel selection only)
    traffic, traffic_src = (measured_traffic(kernel, rows, nnz, n)
                            if os.environ.get("SBLAS_SPMM_VARIANT", "") in ("", "auto", "win6") else (None, None))
    out = {
        "metric": "SpMM GFLOP/s (2*nnz*N/t), CSR x dense N=64, fp64",
        "value": round(value, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "settle_steps": settled,   # untimed steps in front of the warm-up until the clocks have settled (see settle())
        "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not args.matrix else "file",
        "config": {"workload": "SpMM method-1 (dense-B column partition), %s, nnz=%d, N=%d columns per GPU, alpha=beta=1, "
                               "inputs resident in HBM; step = B->row-major staging + row-panel SpMM" % (name, nnz, n),
                   "rows": rows, "cols": cols, "nnz": nnz, "n_cols_per_gpu": n, "parallelism": "method1-colblock x%d" % world},
        "roofline": {"bound": "hbm", "kernel": kernel,
                     "achieved": round(alg / t_roof / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / t_roof / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg, "kernel_ms": round(t_roof * 1e3, 5),
                     "kernel_ms_source": "HIP events around the one launch, 10 steps" if t_dom else "HIP events around stage 2",
                     "stage2_ms": round(t_stage2 * 1e3, 5), "stage2_ms_median": round(float(np.median(s2)), 5),
                     "stage2_ms_min": round(float(np.min(s2)), 5),
                     "staging_kernel_ms": round(t_stage1 * 1e3, 5),
                     "kernel_gflops": round(flops_step / t_roof / 1e9, 1)},
        "hbm_gbs_whole_step": round(alg / (elapsed / args.steps) / 1e9, 1),
        "oracle_check": check,
    }

    # ---- method 2 (row-block A + RCCL merge), informational, N total = ncols ---------------------------------
    if world > 1 and not args.no_method2:
        # informational: an exception here (raised on every rank alike) must not cost the method-1 line above
        try:
            part = S.partition_nnz(rp, world, rank)
            lo, k = part["first_nnz"], part["nnz"]
            rp_i, ci_i, v_i = d(part["rowptr"]), colidx[lo:lo + k].contiguous(), val[lo:lo + k].contiguous()
            m_i = len(part["rowptr"]) - 1
            gen0 = torch.Generator(device="cpu").manual_seed(211)
            B2 = torch.rand(cols * n, dtype=torch.float64, generator=gen0).to(dev)     # replicated B
            C2 = torch.ones(rows * n, dtype=torch.float64, device=dev)
            Ccopy = torch.zeros(rows * n, dtype=torch.float64, device=dev)
            e = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]

            def step2(k=None):
                Ccopy.zero_()                                     # spmm.h:182-183 (zero buffer), on device
                if k is not None: e[k][0].record()
                S.dense_to_rowmajor(cols, n, B2, cols, Bt)
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
            # method-2 result check on rank 0: C2 = 1 + (warmup+steps) * A*B2 on 64 sampled rows
            m2_ok = None
            if rank == 0:
                import oracle_py as O2
                r0 = rows // 2
                ref2 = np.zeros(rows * n)
                O2.spmm_rows(r0, r0 + 64, rows, cols, n, rp, ci, v, B2.cpu().numpy(), ref2, 1.0, 0.0)
                got2 = C2.view(n, rows)[:, r0:r0 + 64].cpu().numpy()
                want2 = 1.0 + (args.warmup + args.steps) * ref2.reshape(n, rows)[:, r0:r0 + 64]   # (method 2 has no settling phase)
                m2_ok = bool(np.allclose(got2, want2, rtol=1e-9, atol=1e-9))
                if not m2_ok:
                    failures.append("method-2 bench result does not match the oracle: max diff %g" % np.abs(got2 - want2).max())
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

            # ---- method 2, fast merge: packed row blocks, all-gather, one scatter + alpha/beta pass ---------
            parts = [S.partition_nnz(rp, world, q) for q in range(world)]
            starts = [p_["start_row"] for p_ in parts]
            nrows = [len(p_["rowptr"]) - 1 for p_ in parts]
            maxblk = max(max(nrows), 1) * n
            mine = torch.zeros(maxblk, dtype=torch.float64, device=dev)           # packed m_i x n block (+ padding)
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
                    dist.all_gather_into_tensor(allb, mine)        # half the bytes of the all-reduce
                else:                                              # rehearsal: gloo on host copies
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
                got3 = C3.view(n, rows)[:, r0:r0 + 64].cpu().numpy()
                m3_ok = bool(np.allclose(got3, want2, rtol=1e-9, atol=1e-9))
                if not m3_ok:
                    failures.append("method-2 (row-block merge) bench result does not match the oracle: max diff %g" % np.abs(got3 - want2).max())
            out["method2_rowblocks"] = {
                "oracle_check": m3_ok, "scaling": "strong", "n_total_cols": n,
                "gflops": round(flops_step * args.steps / el3 / 1e9, 2), "ms_per_step": round(el3 / args.steps * 1e3, 5),
                "ms_spmm": round(float(np.mean([x[0].elapsed_time(x[1]) for x in e3])), 5),
                "ms_allgather": round(float(np.mean([x[1].elapsed_time(x[2]) for x in e3])), 5),
                "ms_merge": round(float(np.mean([x[2].elapsed_time(x[3]) for x in e3])), 5),
                "allgather_payload_bytes_per_rank": maxblk * 8,
                "note": "packed row blocks (beta = 0), torch.distributed all_gather_into_tensor (RCCL), "
                        "sblas_hip_merge_rowblocks_local_f64; the C++ API does the same with RCCL send/recv",
            }
        except Exception as ex:
            out["method2_error"] = repr(ex)

    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(rows, cols, n, rp, ci, v, Bh.numpy(), args.cpu_seconds)
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
