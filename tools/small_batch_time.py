"""Latency of one fit batch at the small end (the reference's own test scale and BASELINE config 1), where the
work per launch is tiny and the ~150 launches of a fit are what is timed."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
from bayesian_ensembling_amd.synthetic import synthetic_ensemble
ctx = L.Context(0)
for (M, N, R, T) in [(10, 24, 3, 2), (8, 200, 8, 2), (8, 200, 8, 500), (32, 512, 8, 2)]:
    data, _ = synthetic_ensemble(M, N, R=R, R_obs=3, seed=1)
    X = np.ascontiguousarray(np.swapaxes(data, 1, 2)); y = data.mean(axis=1); nu = data.var(axis=1)
    dX, dy, dnu = ctx.to_device(X), ctx.to_device(y), ctx.to_device(nu)
    dmu, dcov, dchol = ctx.empty((M, N)), ctx.empty((M, N, N)), ctx.empty((M, N, N))
    dld, dhyp, dbasis = ctx.empty((M,)), ctx.empty((M, 2)), ctx.empty((M, 2, N))
    reps = 20 if T < 100 else 2
    for _ in range(2):
        ctx.gp_fit_batch_dev(M, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.gp_fit_batch_dev(M, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis)
    ctx.sync()
    wall = (time.perf_counter() - t0) / reps * 1e3
    print(f"M={M} N={N} R={R} T={T}: {wall:.3f} ms wall per fit batch -> {M / (wall * 1e-3):.0f} fits/s", flush=True)
    for d in (dX, dy, dnu, dmu, dcov, dchol, dld, dhyp, dbasis):
        d.free()
