"""Can the fit run in fp32 and stay within the 1e-3 relative tolerance BASELINE.json quotes for fp32?

CPU experiment with the oracle's arithmetic (oracle/gp_oracle.py) re-run with every array in float32 (LAPACK spotrf /
strsm / sgemm), against the float64 oracle, on the synthetic workload of bench.py.  Also a mixed variant: float32
products, float64 Cholesky factors.  Result recorded in DESIGN.md section 8 (fp32 mode)."""
import os
import sys

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gp_oracle as go  # noqa: E402


def fit_dtype(X, y, nu, T, dt, chol_dt):
    """gp_oracle.gp_fit with storage/products in `dt` and Cholesky factorisations in `chol_dt`."""
    X, y, nu = X.astype(dt), y.astype(dt), nu.astype(dt)
    N = X.shape[0]
    I = np.eye(N, dtype=dt)
    Lam, th = I.copy(), np.zeros(N, dt)
    u = np.array([go.SOFTPLUS_INV_ONE, go.SOFTPLUS_INV_ONE])
    am, av = np.zeros(2), np.zeros(2)

    def chol(A):
        return np.linalg.cholesky(A.astype(chol_dt)).astype(dt)

    def solve(Lt, B, trans="N"):
        return sla.solve_triangular(Lt, B, lower=True, trans=trans).astype(dt)

    for it in range(T):
        var, ls = go.softplus(u[0]), go.softplus(u[1])
        K0, dK = go.kernel_matrix(X.astype(np.float64), var, ls)     # kernel evaluation itself in fp64 (elementwise)
        K0, dK = K0.astype(dt), dK.astype(dt)
        L = chol(K0 + dt(go.JITTER) * I)
        W = L / np.sqrt(nu)[:, None]
        Lam = dt(0.5) * Lam + dt(0.5) * (I + W.T @ W)
        th = dt(0.5) * th + dt(0.5) * (L.T @ (y / nu))
        C = chol(Lam)
        m = solve(C, solve(C, th), "T")
        Y = solve(C, L.T).T
        LS = solve(C, Y.T, "T").T
        a = L @ m - y
        Lbar = np.tril((np.outer(a, m) + LS) / nu[:, None])
        Phi = np.tril(L.T @ Lbar)
        Phi[np.diag_indices_from(Phi)] *= dt(0.5)
        Wm = solve(L, Phi.T, "T").T
        Kbar = solve(L, Wm, "T")
        g = np.array([float(np.sum(Kbar * K0)) / var, float(np.sum(Kbar * dK))])
        g = g * (1.0 - np.exp(-np.array([var, ls])))
        t = it + 1
        am = 0.9 * am + 0.1 * g
        av = 0.999 * av + 0.001 * g * g
        u = u - 0.01 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * am / (np.sqrt(av) + 1e-7)
    var, ls = go.softplus(u[0]), go.softplus(u[1])
    K0 = go.kernel_matrix(X.astype(np.float64), var, ls)[0].astype(dt)
    L = chol(K0 + dt(go.JITTER) * I)
    A = solve(L, K0)
    mu = A.T @ m
    A2 = solve(C, A)
    cov = K0 - A.T @ A + A2.T @ A2 + np.diag(nu)
    return mu.astype(np.float64), cov.astype(np.float64)


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


if __name__ == "__main__":
    for N in (200, 1000, 2000):
        data, _ = go.synthetic_ensemble(1, N, R=8)
        X, y, nu = go.prologue(data[0])
        mu64, cov64, _ = go.gp_fit(X, y, nu, 2)
        for name, dt, cdt in (("fp32 everywhere", np.float32, np.float32), ("fp32 products, fp64 Cholesky", np.float32, np.float64)):
            try:
                mu, cov = fit_dtype(X, y, nu, 2, dt, cdt)
                try:
                    np.linalg.cholesky(cov)
                    pd = "cov positive definite"
                except np.linalg.LinAlgError:
                    pd = "cov NOT positive definite"
                print(f"N={N:5d} {name:30s}: mu rel err {rel(mu, mu64):.2e}  cov rel err {rel(cov, cov64):.2e}  "
                      f"diag(cov) rel err {rel(np.diag(cov), np.diag(cov64)):.2e}  {pd}", flush=True)
            except np.linalg.LinAlgError as e:
                print(f"N={N:5d} {name:30s}: FAILED ({e})", flush=True)
