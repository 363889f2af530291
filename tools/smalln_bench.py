"""Config-5-like check: many small independent fits (N=86 annual horizon) through the general batched path."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
f32 = "--f32" in sys.argv
dt = np.float32 if f32 else np.float64
N, R, T = 86, 8, 2
rng = np.random.default_rng(0)
for B in (1024, 8192):
    tau = np.linspace(0, 1, N)
    data = 1.5 * tau**2 + 0.12 * rng.standard_normal((B, R, N)).cumsum(axis=2) * 0.3
    X = np.ascontiguousarray(np.swapaxes(data, 1, 2)); y = data.mean(axis=1); nu = data.var(axis=1)
    dX, dy, dnu = ctx.to_device(X, dt), ctx.to_device(y, dt), ctx.to_device(nu, dt)
    dmu, dcov, dchol = ctx.empty((B, N), dt), ctx.empty((B, N, N), dt), ctx.empty((B, N, N), dt)
    dld, dhyp, dbasis = ctx.empty((B,)), ctx.empty((B, 2)), ctx.empty((B, 2, N), dt)
    for rep in range(3):
        ctx.timer_start()
        info = ctx.gp_fit_batch_dev(B, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis, f32=f32)
        ms = ctx.timer_stop()
    es = 4 if f32 else 8
    byts = B * (es * (N * R + 2 * N) + es * (2 * N * N + 3 * N))
    print(f"{'f32' if f32 else 'f64'} B={B} N={N} T={T}: {ms:.1f} ms -> {B / (ms * 1e-3):.0f} fits/s ; algorithmic I/O {byts / 1e6:.0f} MB -> {byts / (ms * 1e-3) / 1e9:.1f} GB/s ; info max {int(info.max())}", flush=True)
    for d in (dX, dy, dnu, dmu, dcov, dchol, dld, dhyp, dbasis):
        d.free()
