"""Phase profile of the fused small-member kernel (csrc/small.hip): cycle-counter stamps of workgroup 0, and the
throughput at config-5 scale.  python tools/small_phases.py [N] [R] [B]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
from bayesian_ensembling_amd.synthetic import synthetic_ensemble
ctx = L.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 86
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
rng = np.random.default_rng(0)
tau = np.linspace(0, 1, N)
data = 1.5 * tau**2 + 0.12 * rng.standard_normal((B, R, N)).cumsum(axis=2) * 0.3
X = np.ascontiguousarray(np.swapaxes(data, 1, 2)); y = data.mean(axis=1); nu = data.var(axis=1)
dX, dy, dnu = ctx.to_device(X), ctx.to_device(y), ctx.to_device(nu)
dmu, dcov, dchol = ctx.empty((B, N)), ctx.empty((B, N, N)), ctx.empty((B, N, N))
dld, dhyp, dbasis = ctx.empty((B,)), ctx.empty((B, 2)), ctx.empty((B, 2, N))
for T in (2, 50):
    for rep in range(3):
        ctx.timer_start()
        info = ctx.gp_fit_batch_dev(B, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis)
        ms = ctx.timer_stop()
    print(f"B={B} N={N} R={R} T={T}: {ms:.2f} ms -> {B / (ms * 1e-3):.0f} fits/s; {ms * 1e3 / (B / 256):.1f} us per fit per CU; info max {int(info.max())}", flush=True)
ctx.small_stamps(True)
if len(sys.argv) > 4 and sys.argv[4] == "reduced":
    dvar = ctx.empty((B, N))
    for rep in range(3):
        ctx.timer_start()
        ctx.gp_fit_batch_reduced_dev(B, N, R, dX, dy, dnu, 2, dmu, dvar, dld, dhyp, dbasis)
        ms = ctx.timer_stop()
    print(f"reduced export, T=2: {ms:.2f} ms -> {B / (ms * 1e-3):.0f} fits/s", flush=True)
else:
    ctx.gp_fit_batch_dev(B, N, R, dX, dy, dnu, 2, dmu, dcov, dchol, dld, dhyp, basis=dbasis)
st = ctx.small_stamps(True, read=True).astype(np.int64)
ws = ctx.small_wave_stamps().astype(np.int64)
ctx.small_stamps(False)
names = ["hypers+kmat", "potrf L (+L^-1, G)", "Lam, theta1", "potrf C (+C^-1, S)", "S, G -> LDS", "m, u", "G S -> Phi", "-", "W = Phi Linv", "Kbar, grad, Adam",
         "(to predict)", "P: kmat+potrf (+L^-1)", "-", "P: mu", "P: cov rows", "P: potrf cov", "P: outputs", "-"]
d = np.diff(st[:18])
clk = 100e6   # s_memtime / readcyclecounter ticks at 100 MHz on gfx950? reported in ticks; print raw and the share
tot_it = st[10] - st[0]
print("phase stamps of workgroup 0 (last member, last iteration), ticks and share:")
for k, nme in enumerate(names):
    if k < len(d):
        print(f"  {nme:20s} {d[k]:9d}  {100.0 * d[k] / max(st[17] - st[0], 1):5.1f} %")
print("iteration total", tot_it, "predict total", st[17] - st[11])




if ws.any():
    print("per-wave stamps (cycles since the wave's first stamp; 0 = not placed):")
    for w in range(8):
        base = ws[w, 0]
        print(f"  wave {w}: " + " ".join(f"{(v - base) if v else 0:7d}" for v in ws[w, :8]) + "  |  " +
              " ".join(f"{(v - ws[w, 8]) if v else 0:7d}" for v in ws[w, 8:16]))
