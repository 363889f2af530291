"""Pins the oracle: the NumPy restatement must reproduce the reference's own pickled fits
(experiments/pre_fit_models/*.pkl, GPDTW1D with n_optim_nits=2500) carried by tests/golden/*.npz
(extracted by oracle/extract_golden.py without importing any reference code)."""
import os

import numpy as np
import pytest
from threadpoolctl import threadpool_limits

from conftest import GOLDEN, golden_files, relerr
from oracle import gp_oracle as go


@pytest.mark.parametrize("fname", golden_files())
def test_oracle_reproduces_reference_fit(fname):
    g = np.load(os.path.join(GOLDEN, fname))
    X, _, nu = go.prologue(g["data"])
    with threadpool_limits(1):     # N <= 165: BLAS threading only adds overhead
        mu, cov, hyp = go.gp_fit(X, g["y_mean"], nu, int(g["n_iters"]))
    # posterior covariance: N(N+1)/2 entries explained by 2 hypers + the recovered target -> pins the
    # kernel form, the natgrad recursion, the Adam constants and ordering, the prediction formula
    assert relerr(cov, g["cov_gold"]) < 1e-6
    assert np.max(np.abs(mu - g["mu_gold"])) < 5e-6
    Lc, ld = go.mvn_factor(g["cov_gold"])
    assert np.max(np.abs(Lc - g["chol_gold"])) < 1e-12          # distrax _scale_tri
    assert abs(ld - float(g["logdet_gold"])) < 1e-9              # TriangularLinear._logdet
    assert relerr(hyp, g["hyp"]) < 1e-6


def test_golden_fixture_inventory():
    files = golden_files()
    assert len(files) >= 4
    shapes = {tuple(np.load(os.path.join(GOLDEN, f))["data"].shape) for f in files}
    assert any(s[1] == 165 for s in shapes) and any(s[1] == 86 for s in shapes)
