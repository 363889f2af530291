"""Phase stamps (shader cycles) of the fused mid-size kernel (csrc/medium.hip), workgroup 0, last member / iteration."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
from oracle import gp_oracle as go
ctx = L.Context(0)
ctx.set_medium_min_batch(1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 165
M = int(sys.argv[2]) if len(sys.argv) > 2 else 20
data, _ = go.synthetic_ensemble(M, N, R=9, seed=1)
X = np.ascontiguousarray(np.swapaxes(data, 1, 2)); y = data.mean(axis=1); nu = data.var(axis=1)
dX, dy, dnu = ctx.to_device(X), ctx.to_device(y), ctx.to_device(nu)
dmu, dcov, dchol = ctx.empty((M, N)), ctx.empty((M, N, N)), ctx.empty((M, N, N))
dld, dhyp, dbasis = ctx.empty((M,)), ctx.empty((M, 2)), ctx.empty((M, 2, N))
fit = lambda T: ctx.gp_fit_batch_dev(M, N, 9, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis)
fit(2)
for T in (2, 100):
    ctx.timer_start(); fit(T); ms = ctx.timer_stop()
    print(f"M={M} N={N} T={T}: {ms:.2f} ms")
ctx.small_stamps(True)
fit(2)
st = ctx.small_stamps(True, read=True).astype(np.int64)
ctx.small_stamps(False)
names = ["hypers+kmat", "potrf L (+L^-1, G)", "Lam, theta1", "potrf C (+C^-1, S)", "m, u", "G S -> Phi", "W = Phi Linv", "Kbar, grad, Adam",
         "(to predict)", "P: kmat+potrf (+L^-1)", "P: mu, cov rows", "P: potrf cov + outputs"]
d = np.diff(st[:13])
for k, nme in enumerate(names):
    print(f"  {nme:24s} {d[k]:9d}")
print("iteration total", st[8] - st[0], "predict total", st[12] - st[9])


