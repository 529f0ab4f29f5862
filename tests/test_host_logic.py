"""CPU-only checks of the product's host side: the C-ABI library loads and exports every declared symbol,
its loader and partitioners agree bit-for-bit with the oracle, and argument validation returns codes
instead of launching anything.  No compute call is made here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import ASH85, GOLDEN, ROOT


def test_library_exports_every_declared_symbol(sblas):
    hdr = open(os.path.join(ROOT, "include", "sblas_hip.h")).read()
    declared = set(re.findall(r"\b(sblas_[A-Za-z0-9_]+)\s*\(", hdr))
    assert declared == set(sblas.EXPORTS), declared ^ set(sblas.EXPORTS)
    L = sblas.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.sblas_hip_version() >= 100


def test_product_never_links_the_oracle():
    """The shipped library must not depend on oracle/ (no CPU fallback path)."""
    import subprocess
    so = os.path.join(ROOT, "s-blas_amd", "lib", "libsblas_hip.so")
    out = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "oracle" not in out and "rccl" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "s-blas_amd")):
        for f in files:
            if f.endswith((".hip", ".cpp", ".h", ".py")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in text and "liboracle" not in text and "sblas_oracle" not in text, f


def test_loader_bit_exact_vs_oracle(sblas, oracle):
    paths = [ASH85] + sorted(os.path.join(GOLDEN, "loader_cases", f) for f in os.listdir(os.path.join(GOLDEN, "loader_cases")))
    for p in paths:
        a, b = sblas.read_mtx(p), oracle.read_mtx(p)
        assert a[:4] == b[:4], p
        for x, y in zip(a[4:], b[4:]):
            assert x.dtype == y.dtype and x.tobytes() == y.tobytes(), p


def test_loader_vs_committed_reference_outputs(sblas):
    with open(os.path.join(GOLDEN, "loader_expected.json")) as f:
        exp = json.load(f)
    for name, e in exp.items():
        if name.startswith("_"):
            continue
        m, n, nnz, sym, rp, ci, v = sblas.read_mtx(os.path.join(GOLDEN, "loader_cases", name + ".mtx"))
        assert (m, n, nnz, sym) == (e["m"], e["n"], e["nnz"], e["symmetric"]), name
        assert rp.tolist() == e["rowptr"] and ci.tolist() == e["colidx"], name
        assert [float(x).hex() for x in v] == e["val"], name


def test_loader_errors(sblas, tmp_path):
    with pytest.raises(sblas.SblasError):
        sblas.read_mtx(str(tmp_path / "missing.mtx"))
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n")   # truncated
    with pytest.raises(sblas.SblasError):
        sblas.read_mtx(str(bad))
    oob = tmp_path / "oob.mtx"
    oob.write_text("%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n")     # row out of range
    with pytest.raises(sblas.SblasError):
        sblas.read_mtx(str(oob))


def test_loader_roundtrip_random(sblas, oracle, tmp_path):
    rng = np.random.default_rng(5)
    for sym in (False, True):
        m = 37
        n = 37 if sym else 23
        nz = 300
        i = rng.integers(1, m + 1, nz)
        j = rng.integers(1, n + 1, nz)
        if sym:
            i, j = np.maximum(i, j), np.minimum(i, j)
        v = rng.standard_normal(nz)
        p = tmp_path / ("r%d.mtx" % sym)
        with open(p, "w") as f:
            f.write("%%%%MatrixMarket matrix coordinate real %s\n%d %d %d\n" % ("symmetric" if sym else "general", m, n, nz))
            for a, b, c in zip(i, j, v):
                f.write("%d %d %.17g\n" % (a, b, c))
        a, b = sblas.read_mtx(str(p)), oracle.read_mtx(str(p))
        assert a[:4] == b[:4]
        for x, y in zip(a[4:], b[4:]):
            assert x.tobytes() == y.tobytes()


def test_loader_binary_sidecar_cache(sblas, oracle, tmp_path, monkeypatch):
    """with SBLAS_CSR_CACHE=1 the parsed arrays are written to <file>.csrbin and served from there while
    the source's size and mtime are unchanged -- bit-identical to the text parse; a changed source or a damaged sidecar
    falls back to parsing (and rewrites it); without the switch nothing is written."""
    import os, shutil, time
    src = os.path.join(os.path.dirname(__file__), "golden", "ash85.mtx")
    p = tmp_path / "a.mtx"
    shutil.copy(src, p)
    ref = oracle.read_mtx(str(p))

    def same(a):
        return a[:4] == ref[:4] and all(x.tobytes() == y.tobytes() for x, y in zip(a[4:], ref[4:]))

    monkeypatch.delenv("SBLAS_CSR_CACHE", raising=False)
    assert same(sblas.read_mtx(str(p))) and not os.path.exists(str(p) + ".csrbin")
    monkeypatch.setenv("SBLAS_CSR_CACHE", "1")
    assert same(sblas.read_mtx(str(p)))                       # parse + write
    bin_path = str(p) + ".csrbin"
    assert os.path.exists(bin_path) and os.path.getsize(bin_path) == 64 + 4 * (86 + 523) + 4 + 8 * 523
    # prove the second load is served by the sidecar: plant a recognisable value in it
    raw = bytearray(open(bin_path, "rb").read())
    off_val = 64 + 4 * (86 + 523) + 4
    raw[off_val:off_val + 8] = np.float64(42.5).tobytes()
    open(bin_path, "wb").write(raw)
    got = sblas.read_mtx(str(p))
    assert got[6][0] == 42.5 and got[4].tobytes() == ref[4].tobytes()
    # a touched source invalidates it
    time.sleep(0.01)
    os.utime(p, None)
    assert same(sblas.read_mtx(str(p)))
    # a truncated sidecar is ignored and rewritten
    open(bin_path, "wb").write(bytes(raw[:100]))
    assert same(sblas.read_mtx(str(p)))
    assert os.path.getsize(bin_path) == len(raw)
    assert same(sblas.read_mtx(str(p)))


def test_find_row_matches_linear_scan(sblas, oracle):
    rng = np.random.default_rng(1)
    lens = rng.integers(0, 5, 200)
    lens[[0, 7, 8, 199]] = 0                       # leading / consecutive / trailing empty rows
    rp = np.zeros(201, np.int32)
    rp[1:] = np.cumsum(lens)
    for idx in list(range(int(rp[-1]))) + [-1, int(rp[-1]), int(rp[-1]) + 5]:
        assert sblas.find_row_of_nnz(rp, idx) == oracle.lib().orc_find_row(rp, 200, idx)


def test_partition_nnz_matches_oracle(sblas, oracle, ash85):
    cases = [(ash85["rowptr"], 85)]
    rng = np.random.default_rng(2)
    for rows in (1, 3, 64, 1000):
        lens = rng.integers(0, 9, rows)
        lens[rng.integers(0, rows)] += 1
        rp = np.zeros(rows + 1, np.int32)
        rp[1:] = np.cumsum(lens)
        cases.append((rp, rows))
    for rp, rows in cases:
        nnz = int(rp[-1])
        for g in (1, 2, 3, 4, 8):
            if (g - 1) * ((nnz + g - 1) // g) >= nnz:
                with pytest.raises(sblas.SblasError):        # a rank would own nothing: refused, not UB
                    for i in range(g):
                        sblas.partition_nnz(rp, g, i)
                continue
            covered = 0
            for i in range(g):
                d = sblas.partition_nnz(rp, g, i)
                s, e, k, reb, avg = oracle.partition_nnz(rp, rows, nnz, g, i, exact=True)
                assert (d["start_row"], d["stop_row"], d["nnz"]) == (s, e, k)
                assert d["rowptr"].tolist() == reb.tolist() and d["first_nnz"] == i * avg
                assert (np.diff(d["rowptr"]) >= 0).all() and d["rowptr"][-1] == k
                covered += k
            assert covered == nnz


def test_partition_tables_known_answers(sblas, ash85):
    with open(os.path.join(GOLDEN, "ash85_golden.json")) as f:
        gold = json.load(f)
    for g_str, table in gold["segments"].items():
        g = int(g_str)
        got = [[d["start_row"], d["stop_row"], d["nnz"]] for d in (sblas.partition_nnz(ash85["rowptr"], g, i) for i in range(g))]
        assert got == table
    for i_str, head in gold["rebased_heads_g4"].items():
        assert sblas.partition_nnz(ash85["rowptr"], 4, int(i_str))["rowptr"][:6].tolist() == head
    for g_str, dims in gold["dense_segments_n64"].items():
        g = int(g_str)
        assert [sblas.partition_dense(64, g, i)[1] for i in range(g)] == dims


def test_partition_dense_matches_oracle_and_never_negative(sblas, oracle):
    for first in (1, 8, 9, 64, 100, 256):
        for g in (1, 2, 3, 4, 8):
            tot = 0
            for i in range(g):
                off, dim = sblas.partition_dense(first, g, i)
                o2, d2 = oracle.partition_dense(first, g, i)
                assert dim >= 0 and off + dim <= first
                if d2 >= 0:                      # the reference goes negative for e.g. 9 columns on 8 GPUs
                    assert (off, dim) == (o2, d2) or d2 == 0
                tot += dim
            assert tot == first


def test_argument_validation_returns_codes_without_a_gpu(sblas):
    L = sblas.lib()
    f = L.sblas_hip_spmm_csr_f64_i32
    one = C.c_void_p(16)                          # never dereferenced: validation fails first
    assert f(-1, None, -1, 4, 0, one, one, one, one, 4, 4, 1.0, 0.0, one, 4, one, 1 << 20) == 1
    assert f(-1, None, 4, 4, 3, None, one, one, one, 4, 4, 1.0, 0.0, one, 4, one, 1 << 20) == 1     # rowptr NULL
    assert f(-1, None, 4, 4, 3, one, one, one, one, 3, 4, 1.0, 0.0, one, 4, one, 1 << 20) == 1      # ldb < cols
    assert f(-1, None, 4, 4, 3, one, one, one, one, 4, 4, 1.0, 0.0, one, 3, one, 1 << 20) == 1      # ldc < rows
    assert f(-1, None, 4, 4, 3, one, one, one, one, 4, 4, 1.0, 0.0, one, 4, None, 0) == 3           # no workspace
    assert f(-1, None, 0, 4, 0, one, None, None, one, 4, 4, 1.0, 0.0, one, 4, None, 0) == 0         # empty: no-op
    assert L.sblas_hip_spmv_csr_f64_i32(-1, None, 4, 4, 3, one, one, one, None, 1.0, 0.0, one) == 1
    assert L.sblas_hip_axpby_f64(-1, None, -5, 1.0, one, 1.0, one) == 1
    assert L.sblas_hip_spmm_ldbt(64) == 64 and L.sblas_hip_spmm_ldbt(65) == 128 and L.sblas_hip_spmm_ldbt(8) == 8
    assert L.sblas_hip_spmm_ldbt(33) == 64 and L.sblas_hip_spmm_ldbt(129) == 256
    assert L.sblas_hip_spmm_ldbt(9) == 16 and L.sblas_hip_spmm_ldbt(16) == 16 and L.sblas_hip_spmm_ldbt(17) == 32 and L.sblas_hip_spmm_ldbt(32) == 32   # the widths method 1 hands a GPU at N = 64
    assert L.sblas_hip_spmm_csr_f64_i32_workspace(10, 100, 5, 64) == 101 * 64 * 8 + 96 + 8192   # Bt + zero row; tail: 16 header ints, 1024 column-range pairs, one span + one class per panel, rounded to 16 bytes
    assert b"workspace" in L.sblas_hip_error_string(3)
    # Queen_4147-sized B at N = 256 (8.5 GB row-major) is walked in 128-column chunks: workspace = one chunk
    assert L.sblas_hip_spmm_csr_f64_i32_workspace(4147110, 4147110, 316548962, 256) == 4147111 * 128 * 8 + (64 + 8192 + ((4147110 + 31) // 32) * 12 + 31) // 16 * 16
    # the stage-2-only entry point cannot chunk: a Bt beyond the 32-bit offset window is refused
    assert L.sblas_hip_spmm_csr_rowmajorB_f64_i32(-1, None, 10, 4147110, 5, one, one, one, one, 256, 256, 1.0, 0.0, one, 10) == 1


def test_python_binding_refuses_cpu_tensors(sblas):
    import torch
    t = torch.zeros(4, dtype=torch.float64)
    with pytest.raises(sblas.SblasError):
        sblas.axpby(4, 1.0, t, 1.0, t)


def test_synthetic_generators(sblas):
    from sblas_amd import synth
    rp, ci, v = synth.banded(500, 40, 100)
    assert rp[-1] == 500 * 40 and len(ci) == rp[-1]
    for r in (0, 1, 99, 250, 499):
        c = ci[rp[r]:rp[r + 1]]
        assert (np.diff(c) > 0).all() and c.min() >= max(0, r - 100) and c.max() <= min(499, r + 100)
    rp2, ci2, v2 = synth.banded(500, 40, 100)
    assert (ci == ci2).all() and (v == v2).all()          # deterministic
    rp, ci, v = synth.random_csr(100, 50, 7, empty_every=10, long_row=(3, 300))
    assert rp[4] - rp[3] == 300 and rp[1] - rp[0] == 0 and ci.max() < 50


def test_host_rand0to1_is_the_reference_initialiser(sblas, oracle):
    """sblas_host_fill_rand0to1 = DenseMatrix's constructor fill (matrix.h:519-528): the golden B values the survey
    captured from the reference, and bit-identical to the oracle's restatement."""
    import json, os
    from conftest import GOLDEN
    g = json.load(open(os.path.join(GOLDEN, "ash85_golden.json")))["spmm_n64_a1_b1"]
    B = sblas.rand0to1(85 * 64)
    assert B[0] == g["B_first"] and B[1] == g["B_second"] and B[-1] == g["B_last"]
    assert (B == oracle.rand0to1(85 * 64)).all()
    assert len(sblas.rand0to1(0)) == 0


@pytest.mark.parametrize("field", ["real", "pattern", "integer", "complex"])
@pytest.mark.parametrize("symmetry", ["general", "symmetric"])
def test_loader_parallel_tokenizer_matches_the_sequential_parse(sblas, tmp_path, monkeypatch, field, symmetry):
    """Large files are tokenised by several threads (chunks cut at whitespace, entries re-aligned by a token count):
    the arrays must be bit-identical to the single-threaded parse, whatever the layout of the token stream -- several
    entries per line, entries broken across lines, tabs, blank lines -- and a bad token must still be an error."""
    rng = np.random.default_rng(7)
    M, N, NZ = 300, 257, 20000
    toks = []
    for _ in range(NZ):
        i = int(rng.integers(1, M + 1))
        j = int(rng.integers(1, (i if symmetry == "symmetric" else N) + 1))
        e = [str(i), str(j)]
        if field == "real":
            e.append(repr(float(rng.standard_normal())))
        elif field == "integer":
            e.append(str(int(rng.integers(-50, 50))))
        elif field == "complex":
            e += ["%.17g" % rng.standard_normal(), "%.3e" % rng.standard_normal()]
        toks += e
    seps = rng.choice([" ", "\n", "\t", "  ", "\n\n", " \n"], len(toks))
    body = "".join(t + s_ for t, s_ in zip(toks, seps))
    path = tmp_path / ("m_%s_%s.mtx" % (field, symmetry))
    path.write_text("%%%%MatrixMarket matrix coordinate %s %s\n%% comment\n%d %d %d\n%s" % (field, symmetry, M, N if symmetry == "general" else M, NZ, body))
    monkeypatch.setenv("SBLAS_LOADER_MIN_BYTES", "0")
    out = {}
    for threads in ("1", "7", "13"):
        monkeypatch.setenv("SBLAS_LOADER_THREADS", threads)
        os.utime(path, None)                                   # new mtime: the library caches its last parse
        out[threads] = sblas.read_mtx(str(path))
    for threads in ("7", "13"):
        assert out[threads][:4] == out["1"][:4]
        for a, b in zip(out[threads][4:], out["1"][4:]):
            assert a.tobytes() == b.tobytes()
    # a malformed token in the middle: error on both paths
    bad = tmp_path / "bad.mtx"
    mid = len(body) // 2
    bad.write_text("%%%%MatrixMarket matrix coordinate %s %s\n%d %d %d\n%s" % (field, symmetry, M, M if symmetry == "symmetric" else N, NZ,
                                                                           body[:mid] + " x7 " + body[mid:]))
    for threads in ("1", "7"):
        monkeypatch.setenv("SBLAS_LOADER_THREADS", threads)
        with pytest.raises(sblas.SblasError):
            sblas.read_mtx(str(bad))


def test_loader_glued_tokens_take_the_sequential_reference_faithful_path(sblas, tmp_path, monkeypatch):
    """fscanf lets a conversion stop inside a whitespace token ("1+2 3.5" reads i = 1, j = 2, value 3.5), so a file
    with such tokens has fewer whitespace tokens than fields and the threaded tokenizer's chunk arithmetic does not apply:
    it must notice (every field has to end at whitespace) and leave the file to the sequential loop -- same arrays as with
    one thread, whatever the thread count."""
    rng = np.random.default_rng(3)
    M, N, NZ = 120, 97, 6000
    entries = []
    for k in range(NZ):
        i, j, x = int(rng.integers(1, M + 1)), int(rng.integers(1, N + 1)), float(rng.standard_normal())
        entries.append(("%d+%d %r" if k % 50 == 7 else "%d %d %r") % (i, j, x))      # every 50th entry: indices glued by a sign
    path = tmp_path / "glued.mtx"
    path.write_text("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n%s\n" % (M, N, NZ, "\n".join(entries)))
    monkeypatch.setenv("SBLAS_LOADER_MIN_BYTES", "0")
    out = {}
    for threads in ("1", "5", "11"):
        monkeypatch.setenv("SBLAS_LOADER_THREADS", threads)
        os.utime(path, None)
        out[threads] = sblas.read_mtx(str(path))
    assert out["1"][2] == NZ
    for threads in ("5", "11"):
        assert out[threads][:4] == out["1"][:4]
        for a, b in zip(out[threads][4:], out["1"][4:]):
            assert a.tobytes() == b.tobytes()


def test_partition_nnz_i64_matches_the_int32_partition(sblas):
    """CsrSparseMatrix<int64_t, T>::sync2gpu(segment) splits exactly as the int32 instantiation does (one template in
    the reference, matrix.h:356-375)."""
    from sblas_amd import synth
    for seed, (rows, avg) in enumerate(((85, 6), (300, 3), (1000, 40))):
        rp, ci, v = synth.random_csr(rows, rows, avg, seed=seed, empty_every=7, long_row=(rows // 3, 5 * avg * 4))
        for g in (1, 2, 3, 4, 8):
            for i in range(g):
                a = sblas.partition_nnz(rp, g, i)
                b = sblas.partition_nnz_i64(rp.astype(np.int64), g, i)
                assert {k: a[k] for k in ("start_row", "stop_row", "nnz", "first_nnz")} == {k: b[k] for k in ("start_row", "stop_row", "nnz", "first_nnz")}
                assert np.array_equal(a["rowptr"].astype(np.int64), b["rowptr"]) and b["rowptr"].dtype == np.int64


def test_typed_oracle_agrees_with_the_fp64_int32_oracle(oracle):
    """The typed restatements (float values, int64 indices) are the same loops: int64 indices give the fp64 results
    bit for bit, fp32 values give the fp64 results to fp32 rounding."""
    from sblas_amd import synth
    rp, ci, v = synth.random_csr(120, 90, 8, seed=4, empty_every=9, long_row=(7, 200))
    rng = np.random.default_rng(0)
    n = 5
    B, C0 = rng.standard_normal(90 * n), rng.standard_normal(120 * n)
    x, y0 = rng.standard_normal(90), rng.standard_normal(120)
    ref = oracle.spmm(120, 90, n, rp, ci, v, B, C0.copy(), 1.5, -0.5)
    refv = oracle.spmv(120, rp, ci, v, x, y0.copy(), 1.5, -0.5)
    rp8, ci8 = rp.astype(np.int64), ci.astype(np.int64)
    assert np.array_equal(oracle.spmm_typed(120, 90, n, rp8, ci8, v, B, C0.copy(), 1.5, -0.5), ref)
    assert np.array_equal(oracle.spmv_typed(120, rp8, ci8, v, x, y0.copy(), 1.5, -0.5), refv)
    f = np.float32
    for r, c in ((rp, ci), (rp8, ci8)):
        got = oracle.spmm_typed(120, 90, n, r, c, v.astype(f), B.astype(f), C0.astype(f), 1.5, -0.5)
        assert got.dtype == f and np.allclose(got, ref, rtol=1e-4, atol=1e-4)
        gotv = oracle.spmv_typed(120, r, c, v.astype(f), x.astype(f), y0.astype(f), 1.5, -0.5)
        assert gotv.dtype == f and np.allclose(gotv, refv, rtol=1e-4, atol=1e-4)


def test_typed_entry_points_validate_without_a_gpu(sblas):
    L = sblas.lib()
    one = (C.c_double * 4)()
    assert L.sblas_hip_spmm_csr(-1, None, 7, 0, 4, 4, 3, one, one, one, one, 4, 2, 1.0, 0.0, one, 4, None, 0) == 1      # unknown value type
    assert L.sblas_hip_spmm_csr(-1, None, 1, 2, 4, 4, 3, one, one, one, one, 4, 2, 1.0, 0.0, one, 4, None, 0) == 1      # unknown index type
    assert L.sblas_hip_spmm_csr(-1, None, 1, 1, 4, 4, 3, one, one, one, one, 4, 2, 1.0, 0.0, one, 4, None, 0) == 3      # no workspace
    assert L.sblas_hip_spmm_csr(-1, None, 1, 0, 0, 4, 0, one, None, None, one, 4, 2, 1.0, 0.0, one, 4, None, 0) == 0    # no rows
    assert L.sblas_hip_spmv_csr(-1, None, 1, 1, 4, 4, 3, one, None, one, one, 1.0, 0.0, one) == 1                      # nnz > 0 without colidx
    assert L.sblas_hip_axpby(-1, None, 5, 4, 1.0, one, 1.0, one) == 1
    assert L.sblas_hip_spmm_csr_workspace(1, 0, 10, 100, 5, 65) == 100 * 128 * 4      # fp32: cols x (n rounded up to 64)
    assert L.sblas_hip_spmm_csr_workspace(0, 1, 10, 100, 5, 64) == 100 * 64 * 8
    assert L.sblas_hip_spmm_csr_workspace(0, 0, 10, 100, 5, 64) == L.sblas_hip_spmm_csr_f64_i32_workspace(10, 100, 5, 64)
