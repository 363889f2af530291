import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from bayesian_ensembling_amd import _lib
from bayesian_ensembling_amd.synthetic import synthetic_ensemble
ctx = _lib.Context(0)
M, N, R, T, R_obs = 32, 2000, 8, 2, 3
data, obs = synthetic_ensemble(M, N, R=R, R_obs=R_obs)
X = np.ascontiguousarray(np.swapaxes(data, 1, 2))
dX, dy, dnu, dobs = ctx.to_device(X), ctx.to_device(data.mean(axis=1)), ctx.to_device(data.var(axis=1)), ctx.to_device(obs)
dmu, dcov, dchol = ctx.empty((M, N)), ctx.empty((M, N, N)), ctx.empty((M, N, N))
dld, dhyp, dbasis = ctx.empty((M,)), ctx.empty((M, 2)), ctx.empty((M, 2, N))
dll, dvar, dw, dbmu, dbvar = ctx.empty((M, N)), ctx.empty((M, N)), ctx.empty((M, N)), ctx.empty((N,)), ctx.empty((N,))
def t(f, *a, **k):
    t0 = time.perf_counter(); r = f(*a, **k); return (time.perf_counter() - t0) * 1e3
for rep in range(3):
    tf = t(ctx.gp_fit_batch_dev, M, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis)
    t1 = t(ctx.loglik_rows_basis_dev, M, N, R_obs, dbasis, dld, dobs, dll)
    t2 = t(ctx.extract_diag_dev, M, N, dcov, dvar)
    t3 = t(ctx.weights_normalise_dev, M, N, dll, 1e-3, dw)
    t4 = t(ctx.barycentre_dev, M, N, dw, dmu, dvar, dbmu, dbvar)
    print("fit %.2f ms | loglik %.3f diag %.3f normalise %.3f barycentre %.3f ms (wall)" % (tf, t1, t2, t3, t4), flush=True)
