"""NumPy prototype of the fused small-N fit kernel (csrc/small.hip): the same phases on the same three lower-packed
16x16-tile buffers, explicit triangular inverses, in-place inversion, bordered chol(cov) - checked against the oracle.
Development tool (run by hand); the HIP kernel is the product, the oracle the checker."""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import gp_oracle as go  # noqa: E402

T16 = 16


def lt(i, j):
    assert j <= i
    return i * (i + 1) // 2 + j


class Buf:
    """lower-packed tile buffer"""

    def __init__(self, NT):
        self.NT = NT
        self.t = np.full((NT * (NT + 1) // 2, 16, 16), np.nan)     # NaN-poisoned: reads of unwritten tiles show up

    def __getitem__(self, ij):
        return self.t[lt(*ij)]

    def __setitem__(self, ij, v):
        self.t[lt(*ij)] = v

    def sym(self, i, k):
        """tile (i,k) of the symmetric matrix stored by its lower tiles"""
        return self.t[lt(i, k)] if k <= i else self.t[lt(k, i)].T

    def dense_lower(self):
        n = self.NT * 16
        A = np.zeros((n, n))
        for i in range(self.NT):
            for j in range(i + 1):
                A[16 * i:16 * i + 16, 16 * j:16 * j + 16] = self[i, j]
        return A


def diag16(A):
    """factor + invert one 16x16 diagonal tile (lower part read)"""
    L = np.linalg.cholesky(np.tril(A) + np.tril(A, -1).T)
    return L, np.linalg.inv(L)


def potrf(A, D, X=None):
    """in-place blocked Cholesky of the lower-packed buffer A; D[k] = inverse of the diagonal tile k.
    With X: the inverse of the factor by forward substitution riding along (X lower-packed, T accumulators in X)."""
    NT = A.NT
    for k in range(NT):
        A[k, k], D[k] = diag16(A[k, k])
        # panel
        for i in range(k + 1, NT):
            A[i, k] = A[i, k] @ D[k].T
        if X is not None:
            X[k, k] = D[k].copy()
            for i in range(k):
                X[k, i] = D[k] @ X[k, i]              # finalise row k: X_ki = inv_kk T_ki
        # update
        for i in range(k + 1, NT):
            for j in range(k + 1, i + 1):
                A[i, j] = A[i, j] - A[i, k] @ A[j, k].T
        if X is not None:
            for j in range(k + 1, NT):
                for i in range(k + 1):
                    prev = X[j, i] if k > i else np.zeros((16, 16))    # first touch of T_ji is at k == i
                    X[j, i] = prev - A[j, k] @ X[k, i]


def invert_in_place(Lb, D):
    """Lb holds L (lower-packed) with diagonal-tile inverses D; on return it holds L^-1.  Columns last to first."""
    NT = Lb.NT
    for j in range(NT - 1, -1, -1):
        new = {}
        for i in range(j + 1, NT):
            Tt = np.zeros((16, 16))                      # T_i^T = sum_k L_kj^T X_ik^T
            for k in range(j + 1, i + 1):
                Tt += Lb[k, j].T @ Lb[i, k].T            # X_ik: already inverted tiles (k > j)
            new[i] = -(Tt.T @ D[j])
        for i, v in new.items():                         # (barrier) all reads of column j done before the writes
            Lb[i, j] = v
        Lb[j, j] = D[j].copy()


def kmat_tiles(buf, X, N, var, ls, kernel, jitter):
    NT = buf.NT
    Np = 16 * NT
    Xs = np.zeros((Np, X.shape[1]))
    Xs[:N] = X / ls
    s = np.sum(Xs * Xs, axis=1)
    r2 = -2.0 * (Xs @ Xs.T) + s[:, None] + s[None, :]
    K, dK = kern(r2, var, ls, kernel)
    pad = np.arange(Np) >= N
    K[pad, :] = 0
    K[:, pad] = 0
    K[pad, pad] = 1.0
    Kj = K + jitter * np.eye(Np) * (~pad)
    for i in range(NT):
        for j in range(i + 1):
            buf[i, j] = Kj[16 * i:16 * i + 16, 16 * j:16 * j + 16].copy()
    return K, dK


def kern(r2, var, ls, kernel):
    if kernel == 2:
        r2c = np.maximum(r2, 0)
        K = var * np.exp(-0.5 * r2c)
        return K, K * r2c / ls
    clamp = r2 <= 1e-36
    r = np.sqrt(np.maximum(r2, 1e-36))
    if kernel == 0:
        e = np.exp(-go.SQRT3 * r)
        K = var * (1 + go.SQRT3 * r) * e
        dK = 3 * var * r * r * e / ls
    else:
        e = np.exp(-go.SQRT5 * r)
        K = var * (1 + go.SQRT5 * r + 5 / 3 * r * r) * e
        dK = 5 / 3 * var * r * r * (1 + go.SQRT5 * r) * e / ls
    return K, np.where(clamp, 0, dK)


def matvec_lower(B, x, trans=False):
    A = B.dense_lower()
    return (A.T if trans else A) @ x


def matvec_sym(B, x):
    A = B.dense_lower()
    A = np.tril(A) + np.tril(A, -1).T
    return A @ x


def fit(X, y_in, nu_in, T, kernel=0, gamma=0.5, lr=0.01, jitter=1e-6):
    N, R = X.shape
    NT = -(-(N + 2) // 16)
    Np = 16 * NT
    y = np.zeros(Np); y[:N] = y_in
    nu = np.ones(Np); nu[:N] = nu_in
    rs = 1 / np.sqrt(nu)
    Lb, Cb, Wb = Buf(NT), Buf(NT), Buf(NT)
    DL, DC = [None] * NT, [None] * NT
    lam = Buf(NT)                                       # "registers"
    for i in range(NT):
        for j in range(i + 1):
            lam[i, j] = np.eye(16) if i == j else np.zeros((16, 16))
    th = np.zeros(Np)
    m = np.zeros(Np)
    u = np.array([go.SOFTPLUS_INV_ONE] * 2)
    am, av, t = np.zeros(2), np.zeros(2), 0
    have_S = False
    for it in range(T):
        var, ls = go.softplus(u[0]), go.softplus(u[1])
        K0, dK = kmat_tiles(Lb, X, N, var, ls, kernel, jitter)             # P1
        potrf(Lb, DL, X=Wb)                                                # P2: L in Lb, L^-1 rides along into Wb
        G = Buf(NT)                                                        # P3 (registers; accumulated behind P2)
        for i in range(NT):
            for j in range(i + 1):
                acc = np.zeros((16, 16))
                for k in range(i, NT):
                    sk = rs[16 * k:16 * k + 16, None]
                    acc += (Lb[k, i] * sk).T @ (Lb[k, j] * sk)
                G[i, j] = acc
                lam[i, j] = (1 - gamma) * lam[i, j] + gamma * ((np.eye(16) if i == j else 0) + acc)
                Cb[i, j] = lam[i, j].copy()
        ly = matvec_lower(Lb, y / nu, trans=True)                          # L^T (y / nu): kept, u needs it
        th = (1 - gamma) * th + gamma * ly
        Lb.t[:] = np.nan                                                   # L is dead from here on
        potrf(Cb, DC, X=Lb)                                                # P4: C in Cb, C^-1 rides along into Lb
        S = Buf(NT)                                                        # P5 (registers, accumulated behind P4)
        for i in range(NT):
            for j in range(i + 1):
                acc = np.zeros((16, 16))
                for k in range(i, NT):
                    acc += Lb[k, i].T @ Lb[k, j]
                S[i, j] = acc
        # P6: m = C^-T (C^-1 theta1) through the inverse FACTOR while it still lies in Lb (round 4; m = S theta1 through the
        # explicit S loses cond(Lam) eps |theta1| in m, which u = G m - L^T (y / nu) multiplies by |G| again: tests/golden/illcond_*)
        m = matvec_lower(Lb, matvec_lower(Lb, th), trans=True)
        Cb.t[:] = S.t
        have_S = True
        Lb.t[:] = G.t                                                      # G -> Lb (C^-1 dead)
        uvec = matvec_sym(Lb, m) - ly                                      # u = L^T D (L m - y) = G m - L^T (y / nu)
        Phi = Buf(NT)                                                      # P7
        for i in range(NT):
            for j in range(i + 1):
                acc = np.zeros((16, 16))
                for k in range(NT):
                    acc += Lb.sym(i, k) @ Cb.sym(k, j)
                r0, c0 = 16 * i, 16 * j
                v = np.outer(uvec[r0:r0 + 16], m[c0:c0 + 16]) + acc
                if i == j:
                    v = np.tril(v)
                    v[np.diag_indices(16)] *= 0.5
                Phi[i, j] = v
        Lb.t[:] = Phi.t                                                    # Phi -> Lb
        Wn = Buf(NT)                                                       # P9: W = Phi Linv (Linv in Wb)
        for i in range(NT):
            for j in range(i + 1):
                acc = np.zeros((16, 16))
                for k in range(j, i + 1):
                    acc += Lb[i, k] @ Wb[k, j]
                Wn[i, j] = acc
        Lb.t[:] = Wn.t                                                     # W -> Lb
        gv = gl = 0.0                                                      # P10
        for i in range(NT):
            for j in range(NT):
                acc = np.zeros((16, 16))
                for k in range(max(i, j), NT):
                    acc += Wb[k, i].T @ Lb[k, j]                           # Kbar = Linv^T W
                rr = np.arange(16 * i, 16 * i + 16)[:, None]
                cc = np.arange(16 * j, 16 * j + 16)[None, :]
                msk = (rr < N) & (cc < N)
                gv += np.sum(acc * K0[16 * i:16 * i + 16, 16 * j:16 * j + 16] * msk)
                gl += np.sum(acc * dK[16 * i:16 * i + 16, 16 * j:16 * j + 16] * msk)
        g = np.array([gv / var * (1 - np.exp(-var)), gl * (1 - np.exp(-ls))])
        t += 1
        am = 0.9 * am + 0.1 * g
        av = 0.999 * av + 0.001 * g * g
        lr_t = lr * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        u = u - lr_t * am / (np.sqrt(av) + 1e-7)
    # ---- prediction ----
    var, ls = go.softplus(u[0]), go.softplus(u[1])
    if not have_S:
        for i in range(NT):
            for j in range(i + 1):
                Cb[i, j] = np.eye(16) if i == j else np.zeros((16, 16))
    kmat_tiles(Lb, X, N, var, ls, kernel, jitter)
    potrf(Lb, DL, X=Wb)                                                    # Lb = L, Wb = Linv (riding along)

    def At(i, k):
        if k < i:
            return Lb[i, k]
        if k == i:
            return np.tril(Lb[i, i]) - jitter * np.tril(Wb[i, i]).T
        return -jitter * Wb[k, i].T

    mu = matvec_lower(Lb, m) - jitter * matvec_lower(Wb, m, trans=True)
    covt = Buf(NT)
    # one product phase (round 4): F = S At^T column by column (registers of the column's two waves, each half of the
    # contraction range), cov(i, j) = sum_l At(i, l) F(l, j) - J^2 (L^-T L^-1)_ij + diag
    for j in range(NT):
        F = [sum(Cb.sym(l, k) @ At(j, k).T for k in range(NT)) for l in range(NT)]
        for i in range(j, NT):
            halves = [sum(At(i, l) @ F[l] for l in range(h * NT // 2, (h + 1) * NT // 2)) for h in (0, 1)]
            q = np.zeros((16, 16))
            for k in range(i, NT):
                q += Wb[k, i].T @ Wb[k, j]
            halves[(i + j) & 1] = halves[(i + j) & 1] - jitter * jitter * q
            acc = halves[0] + halves[1]
            if i == j:
                acc += np.diag(nu[16 * i:16 * i + 16] + jitter)
            covt[i, j] = acc
    cov = covt.dense_lower()
    cov = np.tril(cov) + np.tril(cov, -1).T
    # bordered chol(cov): rows N, N+1 carry (1^T, mu^T) with a huge diagonal
    B = cov.copy()
    B[N:, :] = 0
    B[:, N:] = 0
    B[np.arange(N, Np), np.arange(N, Np)] = 1.0
    B[N, :N] = 1.0
    B[N + 1, :N] = mu[:N]
    B[N, N] = B[N + 1, N + 1] = 1e30
    for i in range(NT):
        for j in range(i + 1):
            Lb[i, j] = B[16 * i:16 * i + 16, 16 * j:16 * j + 16].copy()
    potrf(Lb, DL)
    Lc = Lb.dense_lower()
    chol = np.tril(Lc[:N, :N])
    logdet = np.sum(np.log(np.diag(chol)))
    basis = Lc[N:N + 2, :N].copy()
    return mu[:N], cov[:N, :N], chol, logdet, basis, np.array([var, ls])


if __name__ == "__main__":
    import scipy.linalg as sla
    worst = 0
    for (N, R, T, kernel, seed) in [(86, 8, 2, 0, 1), (24, 3, 2, 0, 2), (62, 5, 3, 1, 3), (7, 3, 2, 0, 4), (30, 4, 0, 0, 5),
                                    (94, 8, 1, 2, 6), (1, 3, 1, 0, 7), (86, 5, 12, 0, 8)]:
        data, _ = go.synthetic_ensemble(1, N, R=R, seed=seed)
        X, y, nu = go.prologue(data[0])
        mu, cov, chol, ld, basis, hyp = fit(X, y, nu, T, kernel)
        mo, co, ho = go.gp_fit(X, y, nu, T, kernel=kernel)
        Lc, ldo = go.mvn_factor(co)
        rel = lambda a, b: np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300)
        errs = [rel(mu, mo), rel(cov, co), rel(chol, Lc), abs(ld - ldo), rel(hyp, ho),
                rel(basis[0], sla.solve_triangular(Lc, np.ones(N), lower=True)),
                rel(basis[1], sla.solve_triangular(Lc, mo, lower=True))]
        print(N, R, T, kernel, ["%.1e" % e for e in errs])
        worst = max(worst, max(errs))
    print("worst", worst)
