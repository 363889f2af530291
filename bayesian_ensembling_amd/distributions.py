"""The three distrax distribution types the reference stores in `Distribution._dist`
(ensembles/data.py:38-39), with the methods its hot path calls: mean(), variance(), covariance(),
stddev(), log_prob(), sample(seed=).  All arithmetic runs through libbayens on the GPU; there is no
CPU evaluation path.

Reference quirk kept on purpose: `Distribution` calls `dist_type(mu, covariance)`
positionally, so for Normal / MultivariateNormalDiag the second argument - a *variance* everywhere in the
reference - is used as the *scale*.
"""
from __future__ import annotations

import numpy as np

from . import _lib


class Normal:
    """dx.Normal(loc, scale): independent univariate normals, batch shape = loc.shape."""

    def __init__(self, loc, scale):
        self.loc = np.asarray(loc, dtype=np.float64)
        self.scale = np.asarray(scale, dtype=np.float64)

    def mean(self):
        return self.loc

    def stddev(self):
        return self.scale

    def variance(self):
        return self.scale ** 2

    def log_prob(self, x):
        """Elementwise log N(x; loc, scale) (weights.py:95-96); x broadcasts against loc on the last axis."""
        x = np.asarray(x, dtype=np.float64)
        flat = np.ascontiguousarray(np.broadcast_to(x, x.shape[:-1] + self.loc.reshape(-1).shape).reshape(-1, self.loc.size))
        ctx = _lib.default_context()
        loc = self.loc.reshape(1, -1)
        scale = self.scale.reshape(1, -1)
        rows = [ctx.normal_loglik_rows(loc, scale, flat[r:r + 1])[0] for r in range(flat.shape[0])]
        return np.stack(rows).reshape(x.shape[:-1] + (self.loc.size,)) if x.ndim > 1 else rows[0]

    def sample(self, seed=0):
        rng = np.random.default_rng(int(seed))
        return self.loc + self.scale * rng.standard_normal(self.loc.shape)


class MultivariateNormalDiag:
    """dx.MultivariateNormalDiag(loc, scale_diag): event shape (N,)."""

    def __init__(self, loc, scale_diag):
        self.loc = np.asarray(loc, dtype=np.float64).reshape(-1)
        self.scale_diag = np.asarray(scale_diag, dtype=np.float64).reshape(-1)

    def mean(self):
        return self.loc

    def stddev(self):
        return self.scale_diag

    def variance(self):
        return self.scale_diag ** 2

    def covariance(self):
        return np.diag(self.scale_diag ** 2)

    def log_prob(self, x):
        """sum_i log N(x_i; loc_i, scale_i); x (..., N) -> (...)."""
        x = np.asarray(x, dtype=np.float64)
        flat = np.ascontiguousarray(np.broadcast_to(x, x.shape[:-1] + (self.loc.size,)).reshape(-1, self.loc.size))
        ctx = _lib.default_context()
        out = np.array([ctx.normal_loglik_rows(self.loc[None], self.scale_diag[None], flat[r:r + 1])[0].sum()
                        for r in range(flat.shape[0])])
        return out.reshape(x.shape[:-1]) if x.ndim > 1 else out[0]

    def sample(self, seed=0):
        rng = np.random.default_rng(int(seed))
        return self.loc + self.scale_diag * rng.standard_normal(self.loc.shape)


class MultivariateNormalFullCovariance:
    """dx.MultivariateNormalFullCovariance(loc, covariance_matrix).

    distrax factors the covariance on construction (Cholesky + log-determinant of the factor); here the
    factor comes from the GPU - either handed over by the fit that produced the covariance, or computed by
    bayens_mvn_factor_batch."""

    def __init__(self, loc, covariance_matrix, _chol=None, _logdet=None, _basis=None):
        self.loc = np.asarray(loc, dtype=np.float64).reshape(-1)
        self._cov = np.asarray(covariance_matrix, dtype=np.float64)
        N = self.loc.size
        assert self._cov.shape == (N, N), "covariance must be (N, N)"
        if _chol is None or _logdet is None or _basis is None:
            chol, logdet, basis = _lib.default_context().mvn_factor_batch(self._cov[None], self.loc[None])
            _chol, _logdet, _basis = chol[0], float(logdet[0]), basis[0]
        self._scale_tri = np.asarray(_chol, dtype=np.float64)
        self._logdet = float(_logdet)
        # (L^-1 1, L^-1 loc): by-products of the bordered factorisation; all a constant-vector log_prob needs
        self._basis = np.asarray(_basis, dtype=np.float64)

    def mean(self):
        return self.loc

    def covariance(self):
        return self._cov

    def variance(self):
        return np.ascontiguousarray(np.diagonal(self._cov))

    def stddev(self):
        return np.sqrt(self.variance())

    def log_prob(self, x):
        """The form the reference uses (weights.py:98-100): x of shape (T, 1) broadcasts against the event
        shape N, i.e. the density is evaluated at the constant vector x_t * 1 for every t -> (T,)."""
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 2 and x.shape[1] == 1:
            ctx = _lib.default_context()
            N = self.loc.size
            out = np.empty(x.shape[0])
            # the kernel takes observation rows of length N; feed the T values in chunks of N
            vals = x[:, 0]
            for s in range(0, vals.size, N):
                chunk = np.zeros(N)
                seg = vals[s:s + N]
                chunk[:seg.size] = seg
                ll = ctx.loglik_rows_basis(self._basis[None], np.array([self._logdet]), chunk[None])
                out[s:s + seg.size] = ll[0, :seg.size]
            return out
        raise NotImplementedError(
            "MultivariateNormalFullCovariance.log_prob: only the (T, 1) broadcast form used by "
            "LogLikelihoodWeight (ensembles/weights.py:98-100) runs on the device in this version")

    def sample(self, seed=0):
        rng = np.random.default_rng(int(seed))
        return self.loc + self._scale_tri @ rng.standard_normal(self.loc.shape)
