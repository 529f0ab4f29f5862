"""Load time of the MatrixMarket path on a generated file of bench size: the reference's own two-pass fscanf
loader (oracle/_ref, compiled from /root/reference where that tree exists), this library's single-pass parser, and the
binary sidecar (SBLAS_CSR_CACHE=1).  CPU only.
  python tools/loader_bench.py [scale of the nd24k-like stand-in, default 1.0] [directory, default /tmp]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-blas_amd", "python"))
import sblas_amd as S
from sblas_amd import synth
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
d = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
path = os.path.join(d, "nd24k_like_%g.mtx" % scale)
rows, (rp, ci, v) = synth.nd24k_like(scale)
nnz = len(ci)
t0 = time.time()
with open(path, "w") as f:
    f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (rows, rows, nnz))
    r = np.repeat(np.arange(1, rows + 1), np.diff(rp))
    for a in range(0, nnz, 1 << 20):
        b = min(nnz, a + (1 << 20))
        f.write("".join("%d %d %.17g\n" % t for t in zip(r[a:b], ci[a:b] + 1, v[a:b])))
print("wrote %s: %.1f MB, %d entries in %.0f s" % (path, os.path.getsize(path) / 1e6, nnz, time.time() - t0), flush=True)
for p in (path + ".csrbin",):
    if os.path.exists(p):
        os.remove(p)
res = {}
sys.path.insert(0, os.path.join(ROOT, "oracle"))
try:                                    # the reference's own two-pass fscanf loader (mmio_info + mmio_data), as a baseline
    import oracle_py as O
    t0 = time.time(); out = O.read_mtx_ref(path); res["reference loader (oracle/_ref, two fscanf passes)"] = time.time() - t0
except Exception as ex:
    print("reference loader not available here:", ex)
os.environ.pop("SBLAS_CSR_CACHE", None)
t0 = time.time(); out = S.read_mtx(path); res["parse (single pass, this library)"] = time.time() - t0
assert out[2] == nnz and (out[4] == rp).all() and (out[5] == ci).all() and (out[6] == v).all()
os.environ["SBLAS_CSR_CACHE"] = "1"
t0 = time.time(); S.read_mtx(path); res["parse + write sidecar"] = time.time() - t0
t0 = time.time(); out = S.read_mtx(path); res["load from sidecar"] = time.time() - t0
assert (out[4] == rp).all() and (out[5] == ci).all() and (out[6] == v).all()
for k, t in res.items():
    print("%-52s %.2f s  (%.0f MB/s of text)" % (k, t, os.path.getsize(path) / 1e6 / t), flush=True)
os.remove(path); os.remove(path + ".csrbin")
