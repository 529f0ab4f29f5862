"""A bounded run of tools/fuzz_parity.py inside the GPU suite: random matrix families, row blocks, leading dimensions,
alpha / beta, kernel-selection switches and value / index types against the oracle (the tool's docstring has the
details; a failing case prints its parameters and `python tools/fuzz_parity.py --seed 3 --only K` replays it)."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_random_cases_match_the_oracle(sblas, oracle, cuda):
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    try:
        for case in range(120):
            assert fuzz.one_case(case, np.random.default_rng([3, case]), cuda, 4000), case
    finally:
        for k in fuzz.SWITCHES:
            os.environ.pop(k, None)
        sblas.reload_env()
