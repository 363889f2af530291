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
