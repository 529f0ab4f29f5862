import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "s-blas_amd", "python"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
ASH85 = os.path.join(GOLDEN, "ash85.mtx")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """The driver runs build() before the suites; when a developer runs pytest straight away, build here."""
    lib = os.path.join(ROOT, "s-blas_amd", "lib", "libsblas_hip.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    _ensure_built()
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def sblas():
    _ensure_built()
    import sblas_amd
    sblas_amd.lib()
    return sblas_amd


@pytest.fixture(scope="session")
def ash85(oracle):
    m, n, nnz, sym, rowptr, colidx, val = oracle.read_mtx(ASH85)
    return dict(m=m, n=n, nnz=nnz, sym=sym, rowptr=rowptr, colidx=colidx, val=val)


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible: the product has no CPU path")
    return torch.device("cuda:0")
