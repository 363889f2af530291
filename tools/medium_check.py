"""Dev check of the fused mid-size kernel (csrc/medium.hip) against the oracle and the general path, plus its timing at the
reference's workload (N = 165, T = 2500)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
from oracle import gp_oracle as go
from threadpoolctl import threadpool_limits
ctx = L.Context(0)
ctx.set_medium_min_batch(10 ** 6 if os.environ.get('BAYENS_NO_MEDIUM') else 1)
rel = lambda a, b: float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
with threadpool_limits(4):
    for (M, N, R, T, k) in [(2, 95, 4, 2, 0), (3, 130, 4, 2, 0), (2, 165, 9, 3, 0), (1, 174, 5, 2, 0), (1, 175, 5, 1, 0), (2, 200, 8, 2, 0),
                            (1, 254, 6, 2, 0), (2, 130, 4, 2, 1), (1, 100, 8, 2, 2), (1, 128, 4, 51, 0), (1, 200, 8, 0, 0)]:
        data, obs = go.synthetic_ensemble(M, N, R=R, seed=N + R)
        X = np.ascontiguousarray(np.swapaxes(data, 1, 2)); y = data.mean(axis=1); nu = data.var(axis=1)
        out = ctx.gp_fit_batch(X, y, nu, T, kernel=k, want_loss=True)
        e = 0.0
        for m in range(M):
            res = go.gp_fit(X[m], y[m], nu[m], T, kernel=k, loss_every=25) if T > 0 else go.gp_fit(X[m], y[m], nu[m], T, kernel=k)
            mo, co, ho = res[:3]
            Lc, ld = go.mvn_factor(co)
            errs = dict(mu=rel(out["mu"][m], mo), cov=rel(out["cov"][m], co), chol=rel(out["chol"][m], Lc), hyp=rel(out["hyp"][m], ho),
                        ld=abs(out["logdet"][m] - ld) / max(1.0, abs(ld)))
            if T > 0:
                errs["loss"] = rel(out["loss"][m], res[3])
            e = max(e, max(errs.values()))
        print(f"M={M} N={N} R={R} T={T} k={k}: worst {e:.2e} {errs} sym {np.array_equal(out['cov'][0], out['cov'][0].T)} info {out['info'].max()}", flush=True)
# timing at the reference's workload
for (M, N, R, T) in [(1, 165, 9, 250), (4, 165, 9, 250), (8, 165, 9, 250), (20, 165, 9, 250), (20, 165, 9, 2500), (64, 165, 9, 250), (128, 165, 9, 250), (256, 165, 9, 250), (512, 165, 9, 250), (20, 100, 9, 250), (20, 250, 9, 250)]:
    data, _ = go.synthetic_ensemble(M, N, R=R, seed=1)
    X = np.ascontiguousarray(np.swapaxes(data, 1, 2)); y = data.mean(axis=1); nu = data.var(axis=1)
    dX, dy, dnu = ctx.to_device(X), ctx.to_device(y), ctx.to_device(nu)
    dmu, dcov, dchol = ctx.empty((M, N)), ctx.empty((M, N, N)), ctx.empty((M, N, N))
    dld, dhyp, dbasis = ctx.empty((M,)), ctx.empty((M, 2)), ctx.empty((M, 2, N))
    ctx.gp_fit_batch_dev(M, N, R, dX, dy, dnu, 2, dmu, dcov, dchol, dld, dhyp, basis=dbasis)
    ctx.timer_start(); ctx.gp_fit_batch_dev(M, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis); ms = ctx.timer_stop()
    print(f"timing M={M} N={N} T={T}: {ms:.1f} ms = {ms / T * 1e3:.1f} us per iteration ({'general' if os.environ.get('BAYENS_NO_MEDIUM') else 'medium'})", flush=True)
