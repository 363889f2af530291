import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from threadpoolctl import threadpool_limits
from bayesian_ensembling_amd import _lib as L
from oracle import gp_oracle as go
ctx = L.Context(0)
def relerr(a, b): return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
g = np.load('tests/golden/gpfit_histssp434_m1.npz')
X, _, nu = go.prologue(g['data']); y = g['y_mean']
with threadpool_limits(1):
    for T in (1, 2, 5, 10, 25, 50, 100, 200, 400, 800):
        o1 = ctx.gp_fit_batch(X[None], y[None], nu[None], T)
        o2 = ctx.gp_fit_batch(X[None], y[None], nu[None], T)
        mo, co, ho = go.gp_fit(X, y, nu, T)
        print('N=165 T=%d: cov %.2e mu %.2e hyp %.2e | rerun identical %s | hyp gpu %s oracle %s' % (
            T, relerr(o1['cov'][0], co), relerr(o1['mu'][0], mo), relerr(o1['hyp'][0], ho),
            np.array_equal(o1['cov'], o2['cov']), o1['hyp'][0], ho), flush=True)
    for fn in ('gpfit_ssp434_m0.npz', 'gpfit_ssp119_m0.npz'):
        g = np.load('tests/golden/' + fn)
        X, _, nu = go.prologue(g['data']); y = g['y_mean']
        for T in (10, 100, 1000):
            o1 = ctx.gp_fit_batch(X[None], y[None], nu[None], T)
            mo, co, ho = go.gp_fit(X, y, nu, T)
            print(fn, 'T=%d: cov %.2e hyp %.2e' % (T, relerr(o1['cov'][0], co), relerr(o1['hyp'][0], ho)), flush=True)
