#!/usr/bin/env python3
"""Copy the rocprofv3 summaries written by tools/profile.sh (gpurun_out/prof) into profiles/ (tracked) and merge the
FETCH_SIZE / WRITE_SIZE passes into profiles/<round>_hbm_traffic.json, the file bench.py reads `roofline.traffic`
from.  Usage: python tools/collect_profiles.py [round-prefix, default r01]"""
import collections, csv, glob, hashlib, json, os, shutil, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = "gpurun_out/prof", "profiles"
os.makedirs(dst, exist_ok=True)
for op in ("spmm", "spmv"):
    f = sorted(glob.glob(f"{src}/{op}/*/*kernel_stats.csv"), key=os.path.getmtime, reverse=True)  # newest run
    if f:
        shutil.copy(f[0], f"{dst}/{rnd}_{op}_bench_kernel_stats.csv")
        for row in csv.DictReader(open(f[0])):
            print(op, row["Name"][:60], row["Calls"], "avg ns", row["AverageNs"])
# per-kernel durations of the same run split by phase: bench.py first runs untimed clock-settling steps (the device
# leaves its idle clocks over ~150 steps), so the all-calls average of --stats mixes cold and settled launches; the
# last 60 launches of a kernel are the timed region (50) + the launcher-event steps (10)
for op in ("spmm", "spmv"):
    f = sorted(glob.glob(f"{src}/{op}/*/*kernel_trace.csv"), key=os.path.getmtime, reverse=True)
    if not f:
        continue
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if "sblas::" in row["Kernel_Name"]:
            per[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(
                (int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    doc = {"source": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py" +
                     (" --op spmv" if op == "spmv" else "") + " --steps 50 --warmup 5 (kernel_trace.csv of the run "
                     "whose --stats summary is %s_%s_bench_kernel_stats.csv)" % (rnd, op), "kernels": {}}
    for k, v in per.items():
        d = [x[1] for x in sorted(v)]
        tail = d[-60:] if op == "spmm" else d[-100:]
        doc["kernels"][k] = {"calls": len(d), "avg_ns_all_calls": sum(d) / len(d), "first_10_avg_ns": sum(d[:10]) / len(d[:10]),
                             "timed_region_calls": len(tail), "timed_region_avg_ns": sum(tail) / len(tail),
                             "min_ns": min(d)}
        print(op, "phases", k[:50], doc["kernels"][k])
    json.dump(doc, open(f"{dst}/{rnd}_{op}_kernel_phases.json", "w"), indent=1)
# the other widths (method 1 on 2 / 4 / 8 GPUs hands a GPU 32 / 16 / 8 of 64 columns; configs 4 / 5 use 128 / 256) and the SpMV shapes
for tag in ("spmm_n8", "spmm_n16", "spmm_n32", "spmm_n128", "spmm_n256", "spmv_queen", "spmv_short"):
    f = sorted(glob.glob(f"{src}/{tag}/*/*kernel_stats.csv"), key=os.path.getmtime, reverse=True)
    if f:
        shutil.copy(f[0], f"{dst}/{rnd}_{tag}_kernel_stats.csv")
    if os.path.exists(f"{src}/{tag}_bench.json") and os.path.getsize(f"{src}/{tag}_bench.json") > 10:
        shutil.copy(f"{src}/{tag}_bench.json", f"{dst}/{rnd}_{tag}_bench.json")
    if os.path.exists(f"{src}/{tag}.txt"):
        shutil.copy(f"{src}/{tag}.txt", f"{dst}/{rnd}_{tag}.txt")
for name in ("bench_default", "bench_spmv"):
    if os.path.exists(f"{src}/{name}.json") and os.path.getsize(f"{src}/{name}.json") > 10:
        shutil.copy(f"{src}/{name}.json", f"{dst}/{rnd}_{name}.json")

CORR = ("gfx950: FETCH_SIZE counts 64 B per 128 B request -> read bytes = 2*FETCH_SIZE (MI355X_MICROARCH.md, HBM); "
        "WRITE_SIZE exact; unit KiB")


def source_sha16():
    """the fingerprint bench.py compares: a traffic figure is only quoted for the kernel sources it was measured on"""
    h = hashlib.sha256()
    for path in sorted(glob.glob("s-blas_amd/csrc/*")):
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def merge(prefix, path, command, workload):
    per = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in sorted(glob.glob(f"{src}/{prefix}_{c}/*/*counter_collection.csv"), key=os.path.getmtime, reverse=True)[:1]:
            agg = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == c:
                    agg[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
            for k, v in agg.items():
                per[k][c] = sum(v) / len(v)
    if not per:
        return
    doc = {"kernels": {}}   # (one run, one source fingerprint: kernels of earlier collections do not linger)
    doc["command"] = command
    doc["source_sha16"] = source_sha16()
    doc.setdefault("workload", workload)
    for k, v in per.items():
        if "sblas::" not in k or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        doc["kernels"][k] = {"FETCH_SIZE_KB_per_launch": v["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": v["WRITE_SIZE"],
                             "hbm_bytes_per_launch_corrected": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024),
                             "correction": CORR}
        print("traffic", k, doc["kernels"][k]["hbm_bytes_per_launch_corrected"])
    json.dump(doc, open(path, "w"), indent=1)


merge("pmc", f"{dst}/{rnd}_hbm_traffic.json",
      "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0",
      {"rows": 72000, "nnz": 28728000, "n": 64})
merge("pmcspmv", f"{dst}/{rnd}_hbm_traffic_spmv.json",
      "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --op spmv --steps 5 --warmup 1",
      {"rows": 72000, "nnz": 28728000, "n": 1})
for n in (8, 16, 32, 128):
    merge("pmcn%d" % n, f"{dst}/{rnd}_hbm_traffic_n{n}.json",
          "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --ncols %d --steps 5 --warmup 1 --cpu-seconds 0 --no-extras --no-settle" % n,
          {"rows": 72000, "nnz": 28728000, "n": n})
