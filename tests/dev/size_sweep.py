"""Robustness sweep on the GPU: odd sizes against the oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesian_ensembling_amd import _lib as L
from oracle import gp_oracle as go
ctx = L.Context(0)
def relerr(a, b): return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
worst = 0
for (M, N, R, T) in [(2, 1, 3, 1), (2, 2, 3, 2), (1, 3, 2, 2), (2, 63, 3, 2), (2, 64, 3, 2), (2, 65, 3, 2), (1, 127, 50, 1), (2, 128, 5, 2),
                     (1, 129, 4, 2), (1, 255, 6, 1), (1, 257, 6, 1), (3, 320, 7, 1), (1, 1000, 8, 1), (1, 1980, 8, 1), (33, 100, 4, 1)]:
    data, obs = go.synthetic_ensemble(M, N, R=R, seed=N * 7 + R)
    X = np.ascontiguousarray(np.swapaxes(data, 1, 2)); y = data.mean(axis=1); nu = data.var(axis=1)
    out = ctx.gp_fit_batch(X, y, nu, T)
    errs = []
    for m in range(min(M, 2)):
        mo, co, ho = go.gp_fit(X[m], y[m], nu[m], T)
        Lc, ld = go.mvn_factor(co)
        errs.append(max(relerr(out["mu"][m], mo), relerr(out["cov"][m], co), relerr(out["chol"][m], Lc), relerr(out["hyp"][m], ho),
                        abs(out["logdet"][m] - ld) / max(1.0, abs(ld))))
    ll = ctx.loglik_rows_basis(out["basis"], out["logdet"], obs)
    ll2 = ctx.loglik_rows(out["mu"], out["chol"], out["logdet"], obs)
    e2 = relerr(ll, ll2)
    worst = max(worst, max(errs), e2)
    print(f"M={M} N={N} R={R} T={T}: fit err {max(errs):.2e}  basis-vs-chol loglik {e2:.2e} info {out['info'].max()}", flush=True)
print("WORST", worst)
