"""Device fit against the oracle with the data scaled by 1e-4 ... 1e4 (the noise variances by the square): the first
iteration's symmetric form inverts diag(nu) + gamma K instead of I + gamma L^T D L - same conditioning, checked here."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
from oracle import gp_oracle as go

ctx = L.Context(0)
rel = lambda a, b: float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
for N in (86, 300):
    for scale in (1e-4, 1e-2, 1.0, 1e2, 1e4):
        data, _ = go.synthetic_ensemble(2, N, R=6, seed=17)
        data = data * scale
        X = np.ascontiguousarray(np.swapaxes(data, 1, 2)); y = data.mean(axis=1); nu = data.var(axis=1)
        for T in (1, 2, 5):
            out = ctx.gp_fit_batch(X, y, nu, T)
            e = {"mu": 0.0, "cov": 0.0, "hyp": 0.0}
            for m in range(2):
                mo, co, ho = go.gp_fit(X[m], y[m], nu[m], T)
                e["mu"] = max(e["mu"], rel(out["mu"][m], mo)); e["cov"] = max(e["cov"], rel(out["cov"][m], co)); e["hyp"] = max(e["hyp"], rel(out["hyp"][m], ho))
            print(f"N={N:4d} scale {scale:7.0e} T={T}: mu {e['mu']:.1e} cov {e['cov']:.1e} hyp {e['hyp']:.1e}  (nu {nu.min():.1e} .. {nu.max():.1e})", flush=True)
