"""Times the stand-alone bayens_loglik_rows (compact factor -> mean log-likelihood rows) at 32 x 2000."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesian_ensembling_amd import _lib

ctx = _lib.Context(0)
M, N, R_obs = 32, 2000, 3
rng = np.random.default_rng(0)
chol = np.tril(rng.standard_normal((M, N, N)) * 0.01) + np.eye(N) * 1.5
mu = rng.standard_normal((M, N)); obs = rng.standard_normal((R_obs, N))
logdet = 2 * np.log(np.diagonal(chol, axis1=1, axis2=2)).sum(axis=1)
d = {k: ctx.to_device(v) for k, v in dict(mu=mu, chol=chol, ld=logdet, obs=obs).items()}
out = ctx.empty((M, N))
for rep in range(4):
    ctx.timer_start(); ctx.loglik_rows_dev(M, N, R_obs, d["mu"], d["chol"], d["ld"], d["obs"], out); ms = ctx.timer_stop()
    print(f"loglik_rows 32x2000 rep {rep}: {ms:.4f} ms -> {M*N*(N+1)/2*8/ms/1e6:.1f} GB/s of the lower triangle")
from scipy.linalg import solve_triangular
ll = out.download()
z = solve_triangular(chol[5], np.stack([np.ones(N), mu[5]], 1), lower=True)
print("finite", np.isfinite(ll).all())
