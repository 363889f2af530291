"""Re-runs ONE case of tests/dev/fuzz_fit.py (seed, case index) and prints every quantity's error with and without the loss
samples: python tools/fuzz_case.py <seed> <case>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
from oracle import gp_oracle as go

seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = L.Context(0)
rel = lambda a, b: float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
for case in range(want + 1):
    M = int(rng.integers(1, 6)); N = int(rng.choice([1, 2, 5, 24, 63, 64, 65, 86, 127, 128, 129, 165, 200, 255, 257, 300, 400, 513]))
    R = int(rng.integers(1, 12)); T = int(rng.choice([0, 1, 2, 3, 5, 26, 30])); kern = int(rng.integers(0, 3))
    if kern != 0 and N > 300:
        kern = 0
    f32 = bool(rng.integers(0, 2)) and N >= 5 and R >= 3 and kern != 2
    s = int(rng.integers(0, 10**6))
print(f"M={M} N={N} R={R} T={T} kernel={kern} f32={f32} data seed {s}")
data, _ = go.synthetic_ensemble(M, N, R=R, seed=s)
X = np.ascontiguousarray(np.swapaxes(data, 1, 2)); y = data.mean(axis=1); nu = data.var(axis=1) + (1e-3 if R == 1 else 0.0)
print("nu min / max", nu.min(), nu.max(), " |y| max", np.abs(y).max())
for want_loss in (True, False):
    out = ctx.gp_fit_batch(X, y, nu, T, kernel=kern, dtype=np.float32 if f32 else np.float64, want_loss=want_loss)
    for m in range(M):
        res = go.gp_fit(X[m], y[m], nu[m], T, kernel=kern, return_state=True)
        mo, co, ho, st = res[0], res[1], res[2], res[-1]
        print(f"  loss={want_loss} member {m}: mu {rel(out['mu'][m].astype(float), mo):.2e} (|mu| {np.abs(mo).max():.2e}) cov {rel(out['cov'][m].astype(float), co):.2e} "
              f"hyp {rel(out['hyp'][m], ho):.2e} cond(Lam) {np.linalg.cond(st.Lam):.2e} hyp {ho}")
