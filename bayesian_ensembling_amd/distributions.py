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
        self._resident = None
        self._loc = np.asarray(loc, dtype=np.float64).reshape(-1)
        self._cov_h = np.asarray(covariance_matrix, dtype=np.float64)
        N = self._loc.size
        assert self._cov_h.shape == (N, N), "covariance must be (N, N)"
        if _chol is None or _logdet is None or _basis is None:
            chol, logdet, basis = _lib.default_context().mvn_factor_batch(self._cov_h[None], self._loc[None])
            _chol, _logdet, _basis = chol[0], float(logdet[0]), basis[0]
        self._chol_h = np.asarray(_chol, dtype=np.float64)
        self._logdet_h = float(_logdet)
        # (L^-1 1, L^-1 loc): by-products of the bordered factorisation; all a constant-vector log_prob needs
        self._basis_h = np.asarray(_basis, dtype=np.float64)
        self._var_h = None

    @classmethod
    def _from_resident(cls, batch, j):
        """Member j of a ResidentBatch (resident.py): nothing is copied; each piece is fetched from HBM when first read."""
        self = cls.__new__(cls)
        self._resident = (batch, j)
        self._loc = self._cov_h = self._chol_h = self._logdet_h = self._basis_h = self._var_h = None
        return self

    def _row(self, attr, buf_name):
        if getattr(self, attr) is None:
            batch, j = self._resident
            setattr(self, attr, batch.row(getattr(batch, buf_name), j))
        return getattr(self, attr)

    @property
    def loc(self):
        return self._row("_loc", "mu64")

    @property
    def _cov(self):
        return self._row("_cov_h", "cov")

    @property
    def _scale_tri(self):
        return self._row("_chol_h", "chol")

    @property
    def _basis(self):
        return self._row("_basis_h", "basis64")

    @property
    def _logdet(self):
        if self._logdet_h is None:
            batch, j = self._resident
            self._logdet_h = float(batch.row(batch.logdet, j))
        return self._logdet_h

    def mean(self):
        return self.loc

    def covariance(self):
        return self._cov

    def variance(self):
        if self._var_h is None:
            if self._resident is not None:
                batch, j = self._resident
                self._var_h = batch.row(batch.var64, j)
            else:
                self._var_h = np.ascontiguousarray(np.diagonal(self._cov_h))
        return self._var_h

    def stddev(self):
        return np.sqrt(self.variance())

    def log_prob(self, x):
        """distrax semantics: x broadcasts against the event shape (N,).  (N,) -> scalar, (..., N) -> (...); the
        (T, 1) form the reference's weight uses (weights.py:98-100) evaluates the density at the constant vector
        x_t * 1 for every t -> (T,) and runs from the basis alone (no pass over the factor)."""
        x = np.asarray(x, dtype=np.float64)
        ctx = _lib.default_context()
        N = self._resident[0].N if self._resident is not None else self._loc.size
        if x.ndim == 2 and x.shape[1] == 1 and N != 1:
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
        if x.shape[-1] != N:
            raise ValueError(f"log_prob: the last axis of x must be the event size {N} (or x of shape (T, 1))")
        flat = np.ascontiguousarray(x.reshape(-1, N))
        if self._resident is not None:          # the factor never leaves HBM
            batch, j = self._resident
            if batch.dtype == np.float64:
                view = _lib.DeviceView(batch.mu.offset(j * N * 8), (1, N)), _lib.DeviceView(batch.chol.offset(j * N * N * 8), (1, N, N)), \
                    _lib.DeviceView(batch.logdet.offset(j * 8), (1,))
                out = ctx.mvn_log_prob(*view, flat, loc=_lib.DEVICE)[0]
                return out.reshape(x.shape[:-1]) if x.ndim > 1 else float(out[0])
        out = ctx.mvn_log_prob(self.loc[None], self._scale_tri[None], np.array([self._logdet]), flat)[0]
        return out.reshape(x.shape[:-1]) if x.ndim > 1 else float(out[0])

    def sample(self, seed=0):
        rng = np.random.default_rng(int(seed))
        return self.loc + self._scale_tri @ rng.standard_normal(self.loc.shape)


class GridMultivariateNormalFullCovariance:
    """A field of INDEPENDENT full-covariance normals over time, one per grid point: the posterior of
    `GPDTW1D(per_gridpoint=True)` for (realisation, time, latitude, longitude) data (BASELINE config 5).  This type has
    no counterpart in the reference (its GPDTW1D refuses more than (realisation, time), ensembles/models.py:171-174);
    it carries what the reference's consumers read of a posterior - mean() and variance() over the flattened
    (time, lat, lon) points (ensembles/ensemble_scheme.py:63-65 flattens space the same way) - while the weights treat
    every grid point as the reference's 1-D path would (weights.py:93-104 per point).  Lives in HBM (resident.ResidentField);
    the time x time covariances are not kept."""

    def __init__(self, field, m):
        self._field = (field, m)
        self._mean_h = self._var_h = None

    def _rows(self, buf):
        f, m = self._field
        # [G][N] on the device -> (time, lat, lon) order of the data, flattened
        return np.ascontiguousarray(f.member_rows(buf, m).T).reshape(-1)

    def mean(self):
        if self._mean_h is None:
            self._mean_h = self._rows(self._field[0].mu)
        return self._mean_h

    def variance(self):
        if self._var_h is None:
            self._var_h = self._rows(self._field[0].var)
        return self._var_h

    def stddev(self):
        return np.sqrt(self.variance())

    def covariance(self):
        raise NotImplementedError("a per-gridpoint field keeps the marginal variances only (the time x time covariance of "
                                  "every grid point is not stored); fit a single grid point with GPDTW1D() to obtain it")

    def log_prob(self, x):
        raise NotImplementedError("log_prob of a per-gridpoint field is evaluated by LogLikelihoodWeight from the resident "
                                  "basis (bayens_loglik_rows_basis_grouped)")

    def sample(self, seed=0):
        rng = np.random.default_rng(int(seed))
        return self.mean() + self.stddev() * rng.standard_normal(self.mean().shape)      # marginal draws
