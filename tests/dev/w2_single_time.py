import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesian_ensembling_amd import _lib as L
from bayesian_ensembling_amd.synthetic import synthetic_ensemble
ctx = L.Context(0)
M, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 2000
data, obs = synthetic_ensemble(M, N, R=8)
X = np.ascontiguousarray(np.swapaxes(data, 1, 2))
out = ctx.gp_fit_batch(X, data.mean(axis=1), data.var(axis=1), 2)
t = time.time(); w, D = ctx.w2_single(out["mu"], chol=out["chol"]); dt = time.time() - t
print(f"w2_single M={M} N={N}: {dt:.2f} s wall (host-staged), {M*(M-1)//2} pairs, sum w = {w.sum():.15f}")
print("w", np.round(w, 5))
if M <= 4:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import gp_oracle as go
    t = time.time(); w_ref, D_ref = go.w2_weights_single(list(out["mu"]), list(out["cov"])); print("oracle %.1f s" % (time.time() - t))
    print("relerr w %.2e  D %.2e" % (np.max(np.abs(w - w_ref) / w_ref), np.max(np.abs(D - D_ref)) / np.max(D_ref)))
