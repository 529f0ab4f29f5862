"""comm.hip's exchange branches (communicator, grouped send / recv with gather offsets, grouped all-reduce, fp64 and
fp32) executed on the one-GPU box against a stand-in for librccl -- test infrastructure under tests/rccl_stub.  Runs
in a child process: the override of the library name must be in the environment before comm.hip resolves RCCL.
What this does not verify: the real RCCL's ABI and xGMI (test_method2_over_real_devices runs wherever 2+ GPUs exist)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_exchange_branches_run_against_the_rccl_stub(sblas, oracle, cuda):
    import __graft_entry__
    stub = __graft_entry__.build_rccl_stub()
    assert os.path.exists(stub)
    env = dict(os.environ)
    env.pop("SBLAS_SPMM_VARIANT", None)
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "comm_stub_driver.py")], capture_output=True, text=True,
                        timeout=600, env=env)
    assert cp.returncode == 0 and "COMM_STUB_OK" in cp.stdout, (cp.stdout[-2000:], cp.stderr[-3000:])


def test_header_layer_method2_and_its_pipeline_through_the_stub(sblas, cuda):
    """sblas_spmm_csr_v2 / sblas_spmv_csr_v1 as the drivers call them, four logical GPUs on the one device, the merge
    going down the exchange path (stub send / recv on the merge streams of the column-tile pipeline)."""
    import __graft_entry__
    stub = __graft_entry__.build_rccl_stub()
    env = dict(os.environ, SBLAS_RCCL_LIB=stub, SBLAS_COMM_FORCE_EXCHANGE="1")
    golden = os.path.join(ROOT, "tests", "golden", "ash85.mtx")
    bin_ = os.path.join(ROOT, "s-blas_amd", "bin")
    for cmd in (["pipeline_test", golden, "300", "4"], ["spmm_test", "2", golden, "64", "3.0", "4.0", "4"],
                ["spmv_test", golden, "3.0", "4.0", "4"]):
        cp = subprocess.run([os.path.join(bin_, cmd[0])] + cmd[1:], capture_output=True, text=True, timeout=600, env=env)
        out = cp.stdout + cp.stderr
        assert cp.returncode == 0 and ("PASS" in out or "Validation = True" in out) and "MISMATCH" not in out, out[-2000:]
    env["SBLAS_MERGE"] = "allreduce"
    cp = subprocess.run([os.path.join(bin_, "spmm_test"), "2", golden, "64", "3.0", "4.0", "4"], capture_output=True, text=True,
                        timeout=600, env=env)
    assert cp.returncode == 0 and "Validation = True" in cp.stdout, (cp.stdout + cp.stderr)[-2000:]
