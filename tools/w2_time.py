"""Times ModelSimilarityWeight(mode='single') (bayens_w2_single: M(M-1)/2 polar iterations) on the resident posteriors of
a 32 x 2000 fit - BASELINE config 3's weighting step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesian_ensembling_amd import _lib
from bayesian_ensembling_amd.synthetic import synthetic_ensemble
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ctx = _lib.Context(0)
R, T = 8, 2
data, obs = synthetic_ensemble(M, N, R=R, R_obs=3)
X = np.ascontiguousarray(np.swapaxes(data, 1, 2))
dX, dy, dnu = ctx.to_device(X), ctx.to_device(data.mean(axis=1)), ctx.to_device(data.var(axis=1))
dmu, dcov, dchol = ctx.empty((M, N)), ctx.empty((M, N, N)), ctx.empty((M, N, N))
dld, dhyp, dbasis = ctx.empty((M,)), ctx.empty((M, 2)), ctx.empty((M, 2, N))
ctx.gp_fit_batch_dev(M, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis, kernel=_lib.MATERN52)
dw, dD = ctx.empty((M,)), ctx.empty((M, M))
for rep in range(2):
    t0 = time.perf_counter()
    ctx.w2_single_dev(M, N, dmu, dchol, dw, dD)
    ctx.sync()
    dt = time.perf_counter() - t0
    pairs = M * (M - 1) // 2
    print(f"w2_single {M} x {N}: {dt*1e3:.1f} ms, {pairs} pairs, {dt*1e3/pairs:.2f} ms per pair", flush=True)
w = dw.download()
print("weights sum", w.sum(), "finite", np.isfinite(w).all())
