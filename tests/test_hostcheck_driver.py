"""CPU check of the host-side sequencing (csrc/driver.cpp): recursive blocked Cholesky / triangular solves,
the natgrad + Adam iteration and the prediction epilogue, run on naive CPU primitives (oracle/hostcheck)
and compared with the NumPy oracle.  The HIP primitives themselves are checked on the GPU (test_gpu_*)."""
import ctypes

import numpy as np
import pytest

from conftest import member_inputs, relerr
from oracle import gp_oracle as go

dp = ctypes.POINTER(ctypes.c_double)
ip = ctypes.POINTER(ctypes.c_int)


def P(a):
    return a.ctypes.data_as(dp)


def hc_fit(lib, X, y, nu, T, kernel=0):
    B, N, R = X.shape
    mu, cov, chol = np.zeros((B, N)), np.zeros((B, N, N)), np.zeros((B, N, N))
    ld, hyp, info = np.zeros(B), np.zeros((B, 2)), np.zeros(B, dtype=np.int32)
    basis = np.zeros((B, 2, N))
    loss = np.full((B, max((T + 24) // 25, 1)), np.nan)
    lib.hostcheck_gp_fit(B, N, R, P(X), P(y), P(nu), kernel, T, ctypes.c_double(0.5), ctypes.c_double(0.01),
                         ctypes.c_double(1e-6), P(mu), P(cov), P(chol), P(ld), P(hyp), info.ctypes.data_as(ip),
                         P(basis), P(loss))
    hc_fit.basis = basis
    hc_fit.loss = loss
    return mu, cov, chol, ld, hyp, info


@pytest.mark.parametrize("M,N,R,T,kernel", [(2, 86, 5, 3, 0), (1, 200, 8, 2, 0), (1, 130, 4, 2, 1), (1, 64, 3, 2, 2),
                                            (1, 65, 3, 1, 0), (1, 24, 3, 2, 0)])
def test_driver_matches_oracle(hostcheck, M, N, R, T, kernel):
    data, _ = go.synthetic_ensemble(M, N, R=R)
    X, y, nu = member_inputs(data)
    mu, cov, chol, ld, hyp, info = hc_fit(hostcheck, X, y, nu, T, kernel)
    assert not info.any()
    for m in range(M):
        mo, co, ho = go.gp_fit(X[m], y[m], nu[m], T, kernel=kernel)
        Lc, l = go.mvn_factor(co)
        assert relerr(mu[m], mo) < 1e-9 and relerr(cov[m], co) < 1e-9
        assert relerr(chol[m], Lc) < 1e-9 and abs(ld[m] - l) < 1e-8 and relerr(hyp[m], ho) < 1e-11
        # the bordered factorisation carries 1 and mu to L_c^-1 1 and L_c^-1 mu
        import scipy.linalg as sla
        assert relerr(hc_fit.basis[m, 0], sla.solve_triangular(Lc, np.ones(N), lower=True)) < 1e-8
        assert relerr(hc_fit.basis[m, 1], sla.solve_triangular(Lc, mo, lower=True)) < 1e-8


@pytest.mark.parametrize("N,T", [(86, 1), (70, 27), (128, 51)])
def test_loss_trace_matches_the_oracle_elbo(hostcheck, N, T):
    """-ELBO after iterations 0, 25, 50 (models.py:211-215) against oracle neg_elbo."""
    data, _ = go.synthetic_ensemble(1, N, R=4)
    X, y, nu = member_inputs(data)
    hc_fit(hostcheck, X, y, nu, T, 0)
    ref = go.gp_fit(X[0], y[0], nu[0], T, loss_every=25)[3]
    assert hc_fit.loss.shape[1] == len(ref) == (T + 24) // 25
    assert np.allclose(hc_fit.loss[0], ref, rtol=1e-10, atol=0)


def test_blocked_potrf_and_trsm(hostcheck):
    rng = np.random.default_rng(0)
    nb, n, m = 2, 192, 128
    A = rng.standard_normal((nb, n, n))
    S = A @ np.swapaxes(A, 1, 2) + n * np.eye(n)
    Lh = S.copy()
    info = np.zeros(nb, dtype=np.int32)
    hostcheck.hostcheck_potrf(nb, n, P(Lh), info.ctypes.data_as(ip))
    ref = np.linalg.cholesky(S)
    assert relerr(Lh, ref) < 1e-13 and not info.any()
    B = rng.standard_normal((nb, m, n))
    for mode, want in [(0, lambda L, Bm: np.linalg.solve(L, Bm.T).T), (1, lambda L, Bm: np.linalg.solve(L.T, Bm.T).T)]:
        X = B.copy()
        hostcheck.hostcheck_trsm(nb, m, n, P(ref), P(X), mode, 0)
        assert relerr(X, np.stack([want(ref[b], B[b]) for b in range(nb)])) < 1e-12
    Bl = np.tril(rng.standard_normal((nb, n, n)))
    X = Bl.copy()
    hostcheck.hostcheck_trsm(nb, n, n, P(ref), P(X), 1, 1)
    want = np.stack([np.linalg.solve(ref[b].T, Bl[b].T).T for b in range(nb)])
    assert relerr(np.tril(X), np.tril(want)) < 1e-12


def test_not_positive_definite_is_reported(hostcheck):
    n = 128
    S = np.eye(n)[None].copy()
    S[0, 70, 70] = -1.0
    info = np.zeros(1, dtype=np.int32)
    hostcheck.hostcheck_potrf(1, n, P(S), info.ctypes.data_as(ip))
    assert info[0] == 71     # 1-based failing pivot


def test_flop_model(hostcheck):
    hostcheck.hostcheck_fit_flops.restype = ctypes.c_double
    f = hostcheck.hostcheck_fit_flops(2000, 8, 2)
    n3 = 2000.0 ** 3
    import bench
    # 13/3 N^3 per iteration, 4/3 N^3 less in the first (G S = (I - S) / gamma from S alone), 3 N^3 prediction, N^3/3 chol(cov)
    # ... and 8/3 N^3 less when another iteration follows it (hyper-gradient from P = (diag(nu) + gamma K)^-1 alone)
    assert abs(f - (2 * (13 / 3 * n3 + 2 * 2000 * 2000 * 8) - 8 / 3 * n3 + 3 * n3 + 2 * 2000 * 2000 * 8 + n3 / 3)) < 1.0
    assert abs(bench.fit_flops(2000, 8, 1) - hostcheck.hostcheck_fit_flops(2000, 8, 1)) < 1.0
    assert abs(bench.fit_flops(2000, 8, 2) - f) < 1.0 and abs(bench.fit_flops(2000, 8, 0) - hostcheck.hostcheck_fit_flops(2000, 8, 0)) < 1.0


def test_polar_iteration_gives_the_w2_trace_term(hostcheck):
    """||L_j^T L_i||_* from the QDWH polar iteration == tr (S_i^1/2 S_j S_i^1/2)^1/2 of wasserstein.py:41-45."""
    rng = np.random.default_rng(0)
    N, nb = 40, 3
    covs = []
    for b in range(nb + 1):
        A = rng.standard_normal((N, N))
        covs.append(A @ A.T / N + 0.01 * (1 + b) * np.eye(N))
    L = np.ascontiguousarray(np.stack([np.linalg.cholesky(c) for c in covs]))
    nuc = np.zeros(nb)
    iters = hostcheck.hostcheck_nuclear(nb, N, P(L), P(nuc))
    ref = [np.trace(go.sqrtm_svd(go.sqrtm_svd(covs[0]) @ covs[j] @ go.sqrtm_svd(covs[0]))) for j in range(1, nb + 1)]
    assert iters <= 8 and relerr(nuc, np.array(ref)) < 1e-12


def test_ragged_batch_through_the_driver_equals_the_members_fitted_alone(hostcheck):
    """driver.cpp with FitParams::Rv / xoff (bayens_gp_fit_batch_ragged): members of R = 3, 7, 4 in one batch - blocks of X at
    explicit offsets, out of order and with a gap - against the same members fitted alone, and one of them against the oracle."""
    N, T, Rs = 70, 3, [3, 7, 4]
    datas = [go.synthetic_ensemble(1, N, R=r, seed=10 + r)[0][0] for r in Rs]          # (R, N) each
    Xs = [np.ascontiguousarray(d.T) for d in datas]
    y = np.stack([d.mean(axis=0) for d in datas]); nu = np.stack([d.var(axis=0) for d in datas])
    # layout: member 2 first, then a gap of 5 doubles, member 0, member 1
    order, off, pos = [2, 0, 1], np.zeros(3, dtype=np.int64), 0
    chunks = []
    for k in order:
        if k == 0:
            chunks.append(np.full(5, np.nan)); pos += 5
        off[k] = pos
        chunks.append(Xs[k].ravel()); pos += Xs[k].size
    X = np.concatenate(chunks)
    B = 3
    mu, cov, chol = np.zeros((B, N)), np.zeros((B, N, N)), np.zeros((B, N, N))
    ld, hyp, info, basis = np.zeros(B), np.zeros((B, 2)), np.zeros(B, dtype=np.int32), np.zeros((B, 2, N))
    Rv = np.array(Rs, dtype=np.int32)
    hostcheck.hostcheck_gp_fit_ragged(B, N, int(Rv.max()), Rv.ctypes.data_as(ip), off.ctypes.data_as(ctypes.POINTER(ctypes.c_long)),
                                      P(X), P(y), P(nu), 0, T, ctypes.c_double(0.5), ctypes.c_double(0.01), ctypes.c_double(1e-6),
                                      P(mu), P(cov), P(chol), P(ld), P(hyp), info.ctypes.data_as(ip), P(basis))
    assert not info.any()
    for m in range(B):
        m1, c1, l1, d1, h1, i1 = hc_fit(hostcheck, Xs[m][None], y[m][None], nu[m][None], T)
        assert np.array_equal(mu[m], m1[0]) and np.array_equal(cov[m], c1[0]) and np.array_equal(chol[m], l1[0])
        assert ld[m] == d1[0] and np.array_equal(hyp[m], h1[0]) and np.array_equal(basis[m], hc_fit.basis[0])
    mo, co, ho = go.gp_fit(Xs[1], y[1], nu[1], T)
    assert relerr(mu[1], mo) < 1e-9 and relerr(cov[1], co) < 1e-9 and relerr(hyp[1], ho) < 1e-11
