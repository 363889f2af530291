"""ensembles/wasserstein.py on the GPU: the 1-D Gaussian Wasserstein barycentre fixed point (:61-100).
The diagonal Gaussian W2 distance (:21-47 with full_cov=False) is fused into the all-pairs kernel behind
ModelSimilarityWeight(mode='temporal')."""
from __future__ import annotations

import warnings

import numpy as np

from . import _lib


def gaussian_barycentre(means, std_devs, weights, tolerance: float = 1e-6, init_var=1.0):
    """wasserstein.py:61-100 for one point: returns (mu, sigma).  Runs the batched kernel with P = 1."""
    means = np.asarray(means, dtype=np.float64).reshape(-1, 1)
    var = np.asarray(std_devs, dtype=np.float64).reshape(-1, 1) ** 2
    w = np.asarray(weights, dtype=np.float64).reshape(-1, 1)
    mu, v, bad = _lib.default_context().barycentre(w, means, var, tolerance, init_var)
    if bad:
        warnings.warn("Barycentre not converged for 1 time step")
    return float(mu[0]), float(np.sqrt(v[0]))


def gaussian_w2_distance_distrax(alpha, beta, full_cov=True):
    """wasserstein.py:21-47 between two of this package's distributions, on the device (bayens_w2_single, M = 2).

    full_cov=True uses covariance() of both (Cholesky factors are taken from the distributions, a diagonal
    covariance has the factor diag(stddev)); full_cov=False uses diag(variance())."""
    from . import distributions as dx
    ctx = _lib.default_context()
    mu = np.stack([np.asarray(alpha.mean(), dtype=np.float64).reshape(-1), np.asarray(beta.mean(), dtype=np.float64).reshape(-1)])
    if not full_cov:
        sd = np.stack([np.sqrt(np.asarray(d.variance(), dtype=np.float64).reshape(-1)) for d in (alpha, beta)])
        _, D = ctx.w2_single(mu, sd=sd)
        return float(D[0, 1])

    def factor(d):
        if isinstance(d, dx.MultivariateNormalFullCovariance):
            return d._scale_tri
        return np.diag(np.asarray(d.stddev(), dtype=np.float64).reshape(-1))
    _, D = ctx.w2_single(mu, chol=np.stack([factor(alpha), factor(beta)]))
    return float(D[0, 1])
