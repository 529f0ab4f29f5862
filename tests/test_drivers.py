"""The command-line drivers (same CLI and report lines as the reference's unit_test / spmm_test / spmv_test) and
the drop-in claim: the reference's own driver sources compile unchanged against s-blas_amd/include."""
import json
import os
import shutil
import subprocess

import pytest

from conftest import ASH85, GOLDEN, ROOT

BIN = os.path.join(ROOT, "s-blas_amd", "bin")
REF = "/root/reference"


def run(exe, *args, cwd=None, env=None):
    p = subprocess.run([os.path.join(BIN, exe)] + [str(a) for a in args], capture_output=True, text=True, cwd=cwd, timeout=600,
                       env=None if env is None else dict(os.environ, **env))
    return p.returncode, p.stdout + p.stderr


def test_config1_cpu_path_reproduces_reference_numbers(sblas):
    """BASELINE config 1: ash85 SpMM method 1, N=64, alpha=beta=1 on the CPU path (gpus = 0, no GPU needed);
    the printed C[0] / C[last] / sum are the reference verifier's own values."""
    with open(os.path.join(GOLDEN, "ash85_golden.json")) as f:
        g = json.load(f)["spmm_n64_a1_b1"]
    rc, out = run("spmm_test", 1, ASH85, 64, 1.0, 1.0, 0)
    assert rc == 0, out
    line = [l for l in out.splitlines() if l.startswith("C[0]")][0]
    vals = [float(tok) for tok in line.replace("=", " ").split() if tok[0].isdigit() or tok[0] == "-"]
    assert vals == [g["C_first"], g["C_last"], g["C_sum"]]
    assert "Validation = True" in out and "input matrix A: ( 85, 85 ) nnz = 523" in out


def test_cli_errors(sblas):
    rc, out = run("spmm_test", 3, ASH85, 64, 1, 1, 0)
    assert rc == 1 and "Method can be only 1 or 2." in out
    rc, out = run("spmm_test")
    assert rc == 1 and "A_path B_width alpha beta gpus" in out
    rc, out = run("spmv_test", "/nonexistent.mtx", 1, 1, 0)
    assert rc != 0


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_reference_drivers_compile_unchanged_against_our_headers(sblas, tmp_path):
    """Drop-in check: the reference's spmm_test.cu / spmv_test.cu / unit_test.cu, byte for byte, built with hipcc
    against s-blas_amd/include + libsblas_hip.so.  The sources are copied to a temp dir OUTSIDE the repo only
    because a quoted #include searches the including file's own directory first."""
    for name in ("spmm_test", "spmv_test", "unit_test"):
        src = tmp_path / (name + ".cu")
        shutil.copyfile(os.path.join(REF, name + ".cu"), src)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-w", "-x", "hip", str(src),
               "-I" + os.path.join(ROOT, "s-blas_amd", "include"), "-I" + os.path.join(ROOT, "include"),
               "-L" + os.path.join(ROOT, "s-blas_amd", "lib"), "-lsblas_hip", "-lpthread", "-o", str(tmp_path / name)]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [1, 2, 4])
def test_drivers_on_gpu(sblas, cuda, gpus):
    """BASELINE config 2 (ash85 SpMV on the GPU vs the CPU verifier) and both SpMM methods through the header API.
    More logical GPUs than devices are folded onto the device (placement + merge logic still run)."""
    for args in (("spmv_test", ASH85, 1.0, 1.0, gpus), ("spmv_test", ASH85, 3.0, 4.0, gpus),
                 ("spmm_test", 1, ASH85, 64, 1.0, 1.0, gpus), ("spmm_test", 2, ASH85, 64, 1.0, 1.0, gpus),
                 ("spmm_test", 2, ASH85, 256, 3.0, 4.0, gpus)):
        rc, out = run(*args)
        assert rc == 0 and "Validation = True" in out, out[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"SBLAS_REPLICATE": "p2p"}, {"SBLAS_MERGE": "allreduce"}])
def test_drivers_placement_and_merge_switches(sblas, cuda, env):
    """SBLAS_REPLICATE=p2p: `replicate` placement as one H2D copy + peer copies (matrix.h:331-355's replacement, N3);
    SBLAS_MERGE=allreduce: the reference's merge pattern.  Four logical GPUs (folded onto the devices present)."""
    for args in (("spmm_test", 1, ASH85, 64, 1.0, 1.0, 4), ("spmm_test", 2, ASH85, 64, 3.0, 4.0, 4), ("spmv_test", ASH85, 3.0, 4.0, 4)):
        rc, out = run(*args, env=env)
        assert rc == 0 and "Validation = True" in out, out[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"SBLAS_PLAN": "0"}])
def test_repeated_calls_run_planned_through_the_header_layer(sblas, cuda, env):
    """An iterative caller: sblas_spmm_csr_v1 / _v2 three times on one matrix.  The first call runs unplanned, the second
    makes a per-GPU plan (kept in the CsrSparseMatrix until its next sync2gpu), the third runs on it; every call is
    checked against the host verifier.  SBLAS_PLAN=0: no plans at all."""
    for width, gpus in ((64, 1), (200, 2), (16, 4), (256, 2)):
        rc, out = run("plan_test", ASH85, width, gpus, 3, env=env)
        assert rc == 0 and "plan_test: PASS" in out and "MISMATCH" not in out, out[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("width,gpus", [(256, 4), (300, 2), (130, 3), (520, 2)])
def test_method2_column_tile_pipeline_is_bit_identical_to_the_serial_form(sblas, cuda, width, gpus):
    """sblas_spmm_csr_v2 pipelines SpMM and merge over 128-column tiles on two streams per GPU (the reference is serial,
    spmm.h:253-265): same bits as the one-piece form, a ragged last tile included, and both match the host verifier."""
    rc, out = run("pipeline_test", ASH85, width, gpus)
    assert rc == 0 and "pipeline_test: PASS" in out and "bit-identical: yes" in out, out[-1500:]


@pytest.mark.gpu
def test_unit_test_driver(sblas, cuda, tmp_path):
    shutil.copyfile(ASH85, tmp_path / "ash85.mtx")       # the driver's hard-coded ./ash85.mtx
    rc, out = run("unit_test", cwd=str(tmp_path))
    assert rc == 0 and "unit_test: PASS" in out and out.count("Validation = True") == 3, out[-2000:]
