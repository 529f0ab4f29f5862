#!/bin/bash
# vendor cross-check (comparison only): dump the bench matrix, run rocSPARSE's SpMM / SpMV on it
set -e
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, "s-blas_amd/python")
import importlib.util
spec = importlib.util.spec_from_file_location("synth", "s-blas_amd/python/sblas_amd/synth.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
rows, (rp, ci, v) = m.nd24k_like()
with open("/tmp/nd24k_like.bin", "wb") as f:
    np.array([rows, rows, len(ci)], np.int64).tofile(f); rp.astype(np.int32).tofile(f); ci.astype(np.int32).tofile(f); v.astype(np.float64).tofile(f)
print("dumped", rows, len(ci))
PY
timeout -k 5 300 tools/rocsparse_compare /tmp/nd24k_like.bin ${1:-64}
