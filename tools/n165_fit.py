"""One collection fit at the reference's workload (20 members x N = 165, n_optim_nits = 250 here) through the default
dispatch (the fused mid-size kernel, csrc/medium.hip) - the command the rocprofv3 passes of tools/make_profiles.sh wrap."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
from bayesian_ensembling_amd.synthetic import synthetic_ensemble
ctx = L.Context(0)
M, N, R, T = 20, 165, 9, int(sys.argv[1]) if len(sys.argv) > 1 else 250
data, _ = synthetic_ensemble(M, N, R=R, R_obs=3, seed=0x165)
X = np.ascontiguousarray(np.swapaxes(data, 1, 2))
dX, dy, dnu = ctx.to_device(X), ctx.to_device(data.mean(axis=1)), ctx.to_device(data.var(axis=1))
dmu, dcov, dchol = ctx.empty((M, N)), ctx.empty((M, N, N)), ctx.empty((M, N, N))
dld, dhyp, dbasis = ctx.empty((M,)), ctx.empty((M, 2)), ctx.empty((M, 2, N))
for rep in range(3):
    ctx.timer_start()
    ctx.gp_fit_batch_dev(M, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis)
    ms = ctx.timer_stop()
print(f"M={M} N={N} T={T}: {ms:.1f} ms, {ms / T * 1e3:.1f} us per iteration")
