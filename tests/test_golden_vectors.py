"""The committed golden vectors (tests/golden/oracle_vectors.npz, made by tests/golden/make_vectors.py): the oracle
reproduces them bit for bit on the CPU, the banks match them on the GPU through the C-ABI."""
import os

import numpy as np
import pytest

import golden_cases
from conftest import record_parity

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))


@pytest.mark.parametrize("case", sorted(golden_cases.CASES))
def test_oracle_reproduces_the_committed_vectors(case):
    for key, arr in golden_cases.CASES[case][0]().items():
        want = GOLD["%s.%s" % (case, key)]
        np.testing.assert_array_equal(np.asarray(arr, np.float32), want, err_msg="%s.%s: the oracle changed" % (case, key))


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(golden_cases.CASES))
def test_banks_match_the_committed_vectors(gpu, case):
    _, run, tol = golden_cases.CASES[case]
    for key, arr in run(gpu).items():
        want = GOLD["%s.%s" % (case, key)]
        assert arr.shape == want.shape, (case, key)
        if tol == 0.0:
            np.testing.assert_array_equal(arr, want, err_msg="%s.%s" % (case, key))
        else:
            peak = max(float(np.abs(want).max()), 1e-6)
            record_parity("golden vectors %s: |gpu - stored oracle output| <= %g peak" % (case, tol),
                          float(np.abs(arr - want).max()) / peak, tol, key=key)
            assert float(np.abs(arr - want).max()) <= tol * peak, (case, key, float(np.abs(arr - want).max()) / peak)
