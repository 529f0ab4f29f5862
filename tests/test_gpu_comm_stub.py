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
