import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesian_ensembling_amd import _lib as L
from oracle import gp_oracle as go
ctx = L.Context(0)
def rel(a, b): return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
for (M, N, R, T, k) in [(2, 64, 4, 1, 0), (2, 86, 5, 2, 0), (2, 200, 8, 2, 0), (1, 500, 8, 2, 1), (1, 1000, 8, 2, 0)]:
    data, _ = go.synthetic_ensemble(M, N, R=R)
    Xs, ys, nus = zip(*[go.prologue(data[m]) for m in range(M)])
    X, y, nu = np.stack(Xs), np.stack(ys), np.stack(nus)
    o32 = ctx.gp_fit_batch(X, y, nu, T, kernel=k, dtype=np.float32, want_loss=True)
    o64 = ctx.gp_fit_batch(X, y, nu, T, kernel=k, want_loss=True)
    print(f"M={M} N={N} R={R} T={T} k={k}: info {o32['info']}  mu {rel(o32['mu'], o64['mu']):.2e} cov {rel(o32['cov'], o64['cov']):.2e} "
          f"chol {rel(o32['chol'], o64['chol']):.2e} logdet {np.max(np.abs(o32['logdet'] - o64['logdet'])):.2e} hyp {rel(o32['hyp'], o64['hyp']):.2e} "
          f"basis {rel(o32['basis'], o64['basis']):.2e} loss {rel(o32['loss'], o64['loss']):.2e}", flush=True)
print("DONE")
