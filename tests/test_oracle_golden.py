"""Pins the oracle: the NumPy restatement must reproduce the reference's own pickled fits
(experiments/pre_fit_models/*.pkl, GPDTW1D with n_optim_nits=2500) carried by tests/golden/*.npz
(extracted by oracle/extract_golden.py without importing any reference code)."""
import os

import numpy as np
import pytest
from threadpoolctl import threadpool_limits

from conftest import GOLDEN, golden_files, golden_tolerances, relerr
from oracle import gp_oracle as go


@pytest.mark.parametrize("fname", golden_files())
def test_oracle_reproduces_reference_fit(fname):
    g = np.load(os.path.join(GOLDEN, fname))
    X, _, nu = go.prologue(g["data"])
    with threadpool_limits(1):     # N <= 165: BLAS threading only adds overhead
        mu, cov, hyp = go.gp_fit(X, g["y_mean"], nu, int(g["n_iters"]))
    # posterior covariance: N(N+1)/2 entries explained by 2 hypers + the recovered target -> pins the
    # kernel form, the natgrad recursion, the Adam constants and ordering, the prediction formula.
    # Bounds per fixture (conftest.golden_tolerances): 1e-6 for the nine members with R >= 6; for the nine chaotic small-R
    # members ten times their measured one-ulp envelope - on THIS machine the oracle reproduces all 18 to <= 5e-7 because the
    # recovered target was fitted through it, but another CPU's BLAS rounds differently.
    tol_cov, tol_mu, tol_hyp = golden_tolerances(g)
    assert relerr(cov, g["cov_gold"]) < tol_cov
    # (the DBA target is RECOVERED through the oracle - SURVEY App. B - so mu_gold is pinned to the convergence of that fixed
    #  point: <= 1e-7 for 14 of the 18 members; ssp119_m5 and ssp460_m2 (R = 5, 4) stop at 6e-6 / 3e-6 because their 2500-step
    #  hyper-parameter trajectory is not a contraction in y.  cov_gold above is the independent pin.)
    assert np.max(np.abs(mu - g["mu_gold"])) < tol_mu
    Lc, ld = go.mvn_factor(g["cov_gold"])
    assert np.max(np.abs(Lc - g["chol_gold"])) < 1e-12          # distrax _scale_tri
    assert abs(ld - float(g["logdet_gold"])) < 1e-9              # TriangularLinear._logdet
    assert relerr(hyp, g["hyp"]) < max(1e-6, tol_hyp if tol_hyp > 1e-5 else 0.0)


def test_golden_fixture_inventory():
    files = golden_files()
    assert len(files) == 18                     # every member of experiments/pre_fit_models/*.pkl (5 pickles)
    chaos = {f: float(np.load(os.path.join(GOLDEN, f))["chaos_cov"]) for f in files}
    R = {f: np.load(os.path.join(GOLDEN, f))["data"].shape[0] for f in files}
    assert all(chaos[f] < 1e-12 for f in files if R[f] >= 6) and all(chaos[f] > 1e-9 for f in files if R[f] <= 5)
    shapes = sorted(tuple(np.load(os.path.join(GOLDEN, f))["data"].shape) for f in files)
    # hist: N = 165 with R in {25, 9, 16} / {25, 9, 10}; ssp: N = 86 with R in {25, 25, 5, 6, 3, 5} + {5, 5, 5} + {5, 5, 4}
    assert shapes == sorted([(25, 165), (9, 165), (16, 165), (25, 165), (9, 165), (10, 165), (25, 86), (25, 86), (5, 86), (6, 86),
                             (3, 86), (5, 86), (5, 86), (5, 86), (5, 86), (5, 86), (5, 86), (4, 86)])
