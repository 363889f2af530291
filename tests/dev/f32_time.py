"""fp32 fit at the headline size beside the fp64 one (device-resident buffers, HIP-event timing)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesian_ensembling_amd import _lib as L
from bayesian_ensembling_amd.synthetic import synthetic_ensemble
ctx = L.Context(0)
M, N, R, T = 32, 2000, 8, 2
data, _ = synthetic_ensemble(M, N, R=R, R_obs=3, seed=0xB4E5)
X = np.ascontiguousarray(np.swapaxes(data, 1, 2)); y = data.mean(axis=1); nu = data.var(axis=1)
res = {}
for name, dt in (("f64", np.float64), ("f32", np.float32)):
    dX, dy, dnu = ctx.to_device(X, dt), ctx.to_device(y, dt), ctx.to_device(nu, dt)
    dmu, dcov, dchol = ctx.empty((M, N), dt), ctx.empty((M, N, N), dt), ctx.empty((M, N, N), dt)
    dld, dhyp, dbasis = ctx.empty((M,)), ctx.empty((M, 2)), ctx.empty((M, 2, N), dt)
    for rep in range(3):
        ctx.timer_start()
        info = ctx.gp_fit_batch_dev(M, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis, f32=(name == "f32"))
        ms = ctx.timer_stop()
    print(f"{name}: {ms:.1f} ms per fit batch -> {M / (ms * 1e-3):.0f} member-fits/s ; info max {int(info.max())}", flush=True)
    res[name] = (dmu.download().astype(np.float64), dcov.download().astype(np.float64), dhyp.download())
    for d in (dX, dy, dnu, dmu, dcov, dchol, dld, dhyp, dbasis):
        d.free()
rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
print("f32 vs f64: mu %.2e cov %.2e hyp %.2e" % (rel(res["f32"][0], res["f64"][0]), rel(res["f32"][1], res["f64"][1]), rel(res["f32"][2], res["f64"][2])))
