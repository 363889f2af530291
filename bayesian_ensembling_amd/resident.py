"""Posteriors that stay in HBM behind the plug-in API.

`GPDTW1D.fit_collection` fits every member of one shape in ONE batched call on resident buffers and hands each
`Distribution` a view (batch, index) instead of host arrays.  What the weighting path needs of a posterior - the
log-likelihood basis, log-determinant, mean and marginal variance rows ([N] each) - is read by the weight / scheme
kernels straight from these buffers; the covariance and its Cholesky factor (2 x 8 N^2 bytes per member, 64 MB at
N = 2000) only cross PCIe when a caller asks for them (`Distribution.covariance`, `_dist._scale_tri`, pickling).
The reference keeps these as host (JAX) arrays inside distrax (ensembles/data.py:38-39); same values, different home.
"""
from __future__ import annotations

import numpy as np


class ResidentBatch:
    """Device outputs of one batched fit: mu / cov / chol / basis in the fit's dtype, float64 row tables
    (mean, variance, basis, logdet) for the weight and scheme kernels."""

    def __init__(self, ctx, B, N, dtype=np.float64):
        self.ctx, self.B, self.N, self.dtype = ctx, B, N, np.dtype(dtype)
        dt = self.dtype
        self.mu, self.cov, self.chol = ctx.empty((B, N), dt), ctx.empty((B, N, N), dt), ctx.empty((B, N, N), dt)
        self.logdet, self.hyp, self.basis = ctx.empty((B,)), ctx.empty((B, 2)), ctx.empty((B, 2, N), dt)
        self.var64 = ctx.empty((B, N))
        if dt == np.float64:
            self.mu64, self.basis64 = self.mu, self.basis
        else:
            self.mu64, self.basis64 = ctx.empty((B, N)), ctx.empty((B, 2, N))
        self._bufs = {id(b): b for b in (self.mu, self.cov, self.chol, self.logdet, self.hyp, self.basis, self.var64,
                                         self.mu64, self.basis64)}

    def finish(self):
        """After the fit: the float64 row tables (diag(cov); widened rows for an fp32 fit)."""
        f32 = self.dtype != np.float64
        self.ctx.extract_diag_dev(self.B, self.N, self.cov, self.var64, f32=f32)
        if f32:
            self.ctx.widen_f32_dev(self.B * self.N, self.mu, self.mu64)
            self.ctx.widen_f32_dev(self.B * 2 * self.N, self.basis, self.basis64)

    def row(self, buf, j):
        return buf.download_rows(j, 1)[0].astype(np.float64, copy=False)

    def free(self):
        for b in self._bufs.values():
            b.free()
        self._bufs = {}

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def resident_tables(ctx, dists, want_chol=False):
    """Device tables of M resident full-covariance posteriors, in member order:
    (basis [M,2,N], logdet [M], mean [M,N], var [M,N], chol [M,N,N] or None, owned buffers to free).
    When the members are exactly one batch in order, these ARE the batch's buffers (no copy); otherwise the rows are
    gathered with device-to-device copies (the factors too when want_chol: they never visit the host).  chol is None
    unless want_chol, and for fp32 fits (the pair kernels take fp64 factors: callers fall back to the host)."""
    views = [d._resident for d in dists]
    batch0 = views[0][0]
    N, M = batch0.N, len(views)
    f64 = all(b.dtype == np.float64 for b, _ in views)
    if all(b is batch0 for b, _ in views) and [j for _, j in views] == list(range(batch0.B)) and M == batch0.B:
        return batch0.basis64, batch0.logdet, batch0.mu64, batch0.var64, (batch0.chol if want_chol and f64 else None), []
    basis, logdet, mean, var = ctx.empty((M, 2, N)), ctx.empty((M,)), ctx.empty((M, N)), ctx.empty((M, N))
    chol = ctx.empty((M, N, N)) if want_chol and f64 else None
    for m, (b, j) in enumerate(views):
        assert b.N == N, "posteriors of different length in one collection"
        basis.copy_rows_from(m, b.basis64, j)
        logdet.copy_rows_from(m, b.logdet, j)
        mean.copy_rows_from(m, b.mu64, j)
        var.copy_rows_from(m, b.var64, j)
        if chol is not None:
            chol.copy_rows_from(m, b.chol, j)
    return basis, logdet, mean, var, chol, [basis, logdet, mean, var] + ([chol] if chol is not None else [])


class ResidentField:
    """Device outputs of a per-gridpoint fit of a whole collection (BASELINE config 5): M members x G grid points of
    independent (realisation, time) posteriors, fitted by `GPDTW1D(per_gridpoint=True)` in batched launches of the fused
    small-member kernel with the REDUCED export - what the weighting path needs of a posterior and nothing else:
    mean / marginal variance [B][N], log-likelihood basis [B][2][N], log-determinant and hyper-parameters per fit,
    b = m * G + g (member-major, so that [M][G*N] is the [model][point] table the weight and scheme kernels take).
    The N x N covariances of the 442 368 fits of config 5 (26 GB) are never formed in HBM, let alone downloaded."""

    def __init__(self, ctx, M, G, N):
        self.ctx, self.M, self.G, self.N, self.B = ctx, M, G, N, M * G
        B = self.B
        self.mu, self.var = ctx.empty((B, N)), ctx.empty((B, N))
        self.logdet, self.hyp, self.basis = ctx.empty((B,)), ctx.empty((B, 2)), ctx.empty((B, 2, N))
        self._bufs = [self.mu, self.var, self.logdet, self.hyp, self.basis]

    def member_rows(self, buf, m):
        """Member m's [G][...] block of a per-fit table as a host array."""
        return buf.download_rows(m * self.G, self.G)

    def free(self):
        for b in self._bufs:
            b.free()
        self._bufs = []

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def field_of(dists):
    """The ResidentField shared by these posteriors when they are its members 0..M-1 in order, else None."""
    views = [getattr(d, "_field", None) for d in dists]
    if any(v is None for v in views):
        return None
    f0 = views[0][0]
    if all(f is f0 for f, _ in views) and [m for _, m in views] == list(range(f0.M)):
        return f0
    return None
