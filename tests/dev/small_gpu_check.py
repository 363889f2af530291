"""Development check of the fused small-member kernel on the GPU box: parity against the oracle on a few shapes,
then throughput at config-5 scale.  (The committed parity tests are tests/test_gpu_small.py.)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesian_ensembling_amd import _lib as L
from oracle import gp_oracle as go
import scipy.linalg as sla

ctx = L.Context(0)
rel = lambda a, b: float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
worst = 0.0
for (M, N, R, T, kernel, seed) in [(2, 86, 8, 2, 0, 1), (3, 24, 3, 2, 0, 2), (2, 62, 5, 3, 1, 3), (2, 7, 3, 2, 0, 4), (1, 30, 4, 0, 0, 5),
                                   (1, 94, 8, 1, 2, 6), (2, 1, 3, 1, 0, 7), (2, 86, 5, 12, 0, 8), (9, 40, 8, 2, 0, 9), (1, 63, 50, 1, 0, 10)]:
    data, _ = go.synthetic_ensemble(M, N, R=R, seed=seed)
    X = np.ascontiguousarray(np.swapaxes(data, 1, 2)); y = data.mean(axis=1); nu = data.var(axis=1)
    try:
        out = ctx.gp_fit_batch(X, y, nu, T, kernel=kernel)
    except L.BayensError as e:
        print("FAILED", M, N, R, T, kernel, e, getattr(e, "info", None)); continue
    errs = np.zeros(7)
    for m in range(M):
        mo, co, ho = go.gp_fit(X[m], y[m], nu[m], T, kernel=kernel)
        Lc, ld = go.mvn_factor(co)
        e = [rel(out["mu"][m], mo), rel(out["cov"][m], co), rel(out["chol"][m], Lc), abs(out["logdet"][m] - ld), rel(out["hyp"][m], ho),
             rel(out["basis"][m, 0], sla.solve_triangular(Lc, np.ones(N), lower=True)), rel(out["basis"][m, 1], sla.solve_triangular(Lc, mo, lower=True))]
        errs = np.maximum(errs, e)
        assert np.array_equal(out["cov"][m], out["cov"][m].T), "cov not symmetric"
        assert not np.triu(out["chol"][m], 1).any(), "chol has entries above the diagonal"
    print(M, N, R, T, kernel, ["%.1e" % v for v in errs], flush=True)
    worst = max(worst, errs.max())
print("worst", worst, flush=True)

N, R, T = 86, 8, 2
rng = np.random.default_rng(0)
for B in (1024, 8192):
    tau = np.linspace(0, 1, N)
    data = 1.5 * tau**2 + 0.12 * rng.standard_normal((B, R, N)).cumsum(axis=2) * 0.3
    X = np.ascontiguousarray(np.swapaxes(data, 1, 2)); y = data.mean(axis=1); nu = data.var(axis=1)
    dX, dy, dnu = ctx.to_device(X), ctx.to_device(y), ctx.to_device(nu)
    dmu, dcov, dchol = ctx.empty((B, N)), ctx.empty((B, N, N)), ctx.empty((B, N, N))
    dld, dhyp, dbasis = ctx.empty((B,)), ctx.empty((B, 2)), ctx.empty((B, 2, N))
    for rep in range(3):
        ctx.timer_start()
        info = ctx.gp_fit_batch_dev(B, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis)
        ms = ctx.timer_stop()
    print(f"f64 B={B} N={N} T={T}: {ms:.2f} ms -> {B / (ms * 1e-3):.0f} fits/s ({ms * 1e3 / B * min(B, 256):.1f} us per fit per CU); info max {int(info.max())}", flush=True)
    if B == 1024:
        for TT in (10, 50):
            ctx.timer_start(); ctx.gp_fit_batch_dev(B, N, R, dX, dy, dnu, TT, dmu, dcov, dchol, dld, dhyp, basis=dbasis); ms2 = ctx.timer_stop()
            print(f"   T={TT}: {ms2:.2f} ms -> {(ms2) / TT * 1e3 / (B / 256):.1f} us per iteration per member-slot", flush=True)
    for d in (dX, dy, dnu, dmu, dcov, dchol, dld, dhyp, dbasis):
        d.free()
