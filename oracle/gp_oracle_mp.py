"""Extended-precision twin of oracle/gp_oracle.py::gp_fit (TEST INFRASTRUCTURE ONLY; mpmath, 50 digits).

Same arithmetic, statement by statement, as `gp_oracle.gp_fit` (models.py:185-220) - kernel matrix,
Cholesky, closed-form natural-gradient step, Cholesky-backward hyper-gradient, TF-Adam, whitened predict_f + diag(y_var) -
evaluated in multi-precision so that it can arbitrate between the fp64 oracle and the device where the two disagree: on
ill-conditioned members (two nearly coincident realisations -> a variance of ~1e-11 at one time step) BOTH are fp64
evaluations of a problem whose answer fp64 does not determine, and "device vs oracle" then measures nothing.  Only small
cases (N of a few dozen; O(T N^3) Python-level operations).

Used by oracle/make_illcond_golden.py to write tests/golden/illcond_*.npz (inputs, the 50-digit answer rounded to fp64, the
fp64 oracle's own error against it).
"""
from __future__ import annotations

import mpmath as mp

from . import gp_oracle as go


def _chol(A, n):
    L = mp.zeros(n, n)
    for j in range(n):
        s = A[j, j] - mp.fsum(L[j, k] * L[j, k] for k in range(j))
        if s <= 0:
            raise ArithmeticError(f"not positive definite at pivot {j + 1}")
        L[j, j] = mp.sqrt(s)
        for i in range(j + 1, n):
            L[i, j] = (A[i, j] - mp.fsum(L[i, k] * L[j, k] for k in range(j))) / L[j, j]
    return L


def _tri_inv(L, n):
    """inverse of a lower-triangular matrix"""
    X = mp.zeros(n, n)
    for j in range(n):
        X[j, j] = 1 / L[j, j]
        for i in range(j + 1, n):
            X[i, j] = -mp.fsum(L[i, k] * X[k, j] for k in range(j, i)) / L[i, i]
    return X


def _kernel(X, n, R, variance, ls, kernel):
    """K0 and dK/dl   [gp_oracle.kernel_matrix; GPflow Matern32.K as called from models.py:186-189]"""
    K, dK = mp.zeros(n, n), mp.zeros(n, n)
    Xs = [[X[i][d] / ls for d in range(R)] for i in range(n)]
    s = [mp.fsum(v * v for v in row) for row in Xs]
    s3, s5 = mp.sqrt(3), mp.sqrt(5)
    for i in range(n):
        for j in range(n):
            r2 = -2 * mp.fsum(Xs[i][d] * Xs[j][d] for d in range(R)) + s[i] + s[j]
            if kernel == go.KERNEL_RBF:
                r2 = max(r2, mp.mpf(0))
                K[i, j] = variance * mp.exp(-r2 / 2)
                dK[i, j] = K[i, j] * r2 / ls
                continue
            clamp = r2 <= mp.mpf("1e-36")
            r = mp.sqrt(max(r2, mp.mpf("1e-36")))
            if kernel == go.KERNEL_MATERN32:
                e = mp.exp(-s3 * r)
                K[i, j] = variance * (1 + s3 * r) * e
                dK[i, j] = 0 if clamp else 3 * variance * r * r * e / ls
            else:
                e = mp.exp(-s5 * r)
                K[i, j] = variance * (1 + s5 * r + mp.mpf(5) / 3 * r * r) * e
                dK[i, j] = 0 if clamp else mp.mpf(5) / 3 * variance * r * r * (1 + s5 * r) * e / ls
    return K, dK


def _softplus(u):
    return mp.log(1 + mp.exp(u))


def gp_fit_mp(X, y_mean, y_var, n_iters, kernel=go.KERNEL_MATERN32, gamma=0.5, lr=0.01, jitter=go.JITTER, dps=50):
    """(X (N,R), y_mean (N,), y_var (N,)) float64 -> (mu, cov, hyp) as mpmath objects.  The fp64 INPUTS are taken exactly
    (every double is a 50-digit number); constants (gamma, lr, jitter, Adam's betas and epsilon) are the decimal values the
    reference writes, converted from their fp64 representations as TF / GPflow hold them."""
    mp.mp.dps = dps
    n, R = len(X), len(X[0])
    Xm = [[mp.mpf(float(v)) for v in row] for row in X]
    y = [mp.mpf(float(v)) for v in y_mean]
    nu = [mp.mpf(float(v)) for v in y_var]
    g_, lr_, jit = mp.mpf(float(gamma)), mp.mpf(float(lr)), mp.mpf(float(jitter))
    b1, b2, eps = mp.mpf(0.9), mp.mpf(0.999), mp.mpf(1e-7)
    u = [mp.log(mp.e - 1), mp.log(mp.e - 1)]
    am, av = [mp.mpf(0)] * 2, [mp.mpf(0)] * 2
    Lam = mp.eye(n)
    th = [mp.mpf(0)] * n
    m = [mp.mpf(0)] * n
    Cinv = mp.eye(n)
    for it in range(n_iters):
        variance, ls = _softplus(u[0]), _softplus(u[1])
        K0, dK = _kernel(Xm, n, R, variance, ls, kernel)
        L = _chol(K0 + jit * mp.eye(n), n)
        # natgrad step [gp_oracle.natgrad_step]
        G = mp.zeros(n, n)
        for i in range(n):
            for j in range(i + 1):
                G[i, j] = G[j, i] = mp.fsum(L[k, i] * L[k, j] / nu[k] for k in range(i, n))
        Lam = (1 - g_) * Lam + g_ * (mp.eye(n) + G)
        Lty = [mp.fsum(L[k, i] * y[k] / nu[k] for k in range(i, n)) for i in range(n)]
        th = [(1 - g_) * th[i] + g_ * Lty[i] for i in range(n)]
        C = _chol(Lam, n)
        Cinv = _tri_inv(C, n)
        S = Cinv.T * Cinv
        m = [mp.fsum(S[i, k] * th[k] for k in range(n)) for i in range(n)]
        # hyper-gradient [gp_oracle.hyper_gradient]
        LS = L * S
        Lm = [mp.fsum(L[i, k] * m[k] for k in range(i + 1)) for i in range(n)]
        a = [Lm[i] - y[i] for i in range(n)]
        Lbar = mp.zeros(n, n)
        for i in range(n):
            for j in range(i + 1):
                Lbar[i, j] = (a[i] * m[j] + LS[i, j]) / nu[i]
        P = L.T * Lbar
        Phi = mp.zeros(n, n)
        for i in range(n):
            for j in range(i):
                Phi[i, j] = P[i, j]
            Phi[i, i] = P[i, i] / 2
        Linv = _tri_inv(L, n)
        Kbar = Linv.T * Phi * Linv
        g_var = mp.fsum(Kbar[i, j] * K0[i, j] for i in range(n) for j in range(n)) / variance
        g_ls = mp.fsum(Kbar[i, j] * dK[i, j] for i in range(n) for j in range(n))
        gu = [g_var * (1 - mp.exp(-variance)), g_ls * (1 - mp.exp(-ls))]
        # Adam [gp_oracle.adam_step]
        t = it + 1
        lr_t = lr_ * mp.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        for k in range(2):
            am[k] = b1 * am[k] + (1 - b1) * gu[k]
            av[k] = b2 * av[k] + (1 - b2) * gu[k] * gu[k]
            u[k] = u[k] - lr_t * am[k] / (mp.sqrt(av[k]) + eps)
    variance, ls = _softplus(u[0]), _softplus(u[1])
    K0, _ = _kernel(Xm, n, R, variance, ls, kernel)
    L = _chol(K0 + jit * mp.eye(n), n)
    A = _tri_inv(L, n) * K0
    mu = [mp.fsum(A[k, i] * m[k] for k in range(n)) for i in range(n)]
    A2 = Cinv * A
    cov = K0 - A.T * A + A2.T * A2
    for i in range(n):
        cov[i, i] += nu[i]
    return mu, cov, [variance, ls], dict(cond_Lam=None)


def to_f64(v):
    import numpy as np
    if isinstance(v, mp.matrix):
        return np.array([[float(v[i, j]) for j in range(v.cols)] for i in range(v.rows)])
    return np.array([float(x) for x in v])
