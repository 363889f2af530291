import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from bayesian_ensembling_amd import _lib as L
from oracle import gp_oracle as go
ctx = L.Context(0)
Rs = [25, 9, 16, 10, 3, 50] * 4
datas = [go.synthetic_ensemble(1, 165, R=r, seed=7 * k + r)[0][0] for k, r in enumerate(Rs)]
Xs = [np.ascontiguousarray(d.T) for d in datas]
y = np.stack([d.mean(axis=0) for d in datas]); nu = np.stack([d.var(axis=0) for d in datas])
out = ctx.gp_fit_batch_ragged(Xs, y, nu, 3, want_loss=True)
print("ragged 24 x 165, R =", sorted(set(Rs)), "T = 3 with loss samples: info", out["info"].max(), "finite", bool(np.isfinite(out["cov"]).all()))
red = ctx.gp_fit_batch_reduced(np.stack([Xs[1], Xs[7]]), y[[1, 7]], nu[[1, 7]], 2)
print("reduced export 2 x 165: finite", bool(np.isfinite(red["var"]).all()))
