"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the GP-fit -> weights -> weighted-posterior path.

This file is a NumPy fp64 restatement of the arithmetic the reference performs on
its hot path.  It is *not* part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
The product (``bayesian_ensembling_amd``) never imports anything under ``oracle/``.

Parity pin: the restatement is checked against the reference's own pickled fits
(``/root/reference/experiments/pre_fit_models/*.pkl``, 18 members, T=2500) through
the fixtures in ``tests/golden/`` that ``oracle/extract_golden.py`` produced
(posterior mean / covariance / Cholesky / logdet).  See ``tests/test_oracle_golden.py``.
The DTW-barycentre target (tslearn, third party, absent) is outside the boundary:
the fixtures carry the ``y_mean`` recovered from the reference's pickled posterior.

Third-party arithmetic restated here (source not under /root/reference):
  GPflow 2.1.5 (Matern32.K, VGP whitened ELBO, NaturalGradient(gamma) XiNat step,
  conditional(white=True, full_cov=True)), TensorFlow 2.8.1 Keras Adam,
  distrax 0.1.2 MultivariateNormalFullCovariance / Normal / MultivariateNormalDiag.

Every function cites the reference file:line it follows.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla

SQRT3 = math.sqrt(3.0)
SQRT5 = math.sqrt(5.0)
LOG2PI = math.log(2.0 * math.pi)

KERNEL_MATERN32 = 0   # reference kernel (models.py:186)
KERNEL_MATERN52 = 1   # extension, no reference call site
KERNEL_RBF = 2        # extension, no reference call site

JITTER = 1e-6         # GPflow default_jitter()
SOFTPLUS_INV_ONE = math.log(math.e - 1.0)   # unconstrained value whose softplus is 1.0


def softplus(u):
    return np.logaddexp(0.0, u)


# --------------------------------------------------------------------------------------
# Row 1: prologue  (models.py:175-182)
# --------------------------------------------------------------------------------------
def prologue(realisation_set, y_mean=None):
    """realisation_set (R,N) -> X (N,R), y_mean (N,), y_var (N,)   [models.py:175-182].

    ``y_mean`` is the DTW barycentre in the reference (models.py:176-178, tslearn); it is a
    boundary *input* here.  When omitted the documented stand-in mean over realisations is used.
    """
    realisation_set = np.asarray(realisation_set, dtype=np.float64)
    y_var = np.var(realisation_set, axis=0)                     # ddof=0, models.py:179
    X = np.ascontiguousarray(realisation_set.T)                  # models.py:182
    if y_mean is None:
        y_mean = realisation_set.mean(axis=0)
    return X, np.asarray(y_mean, dtype=np.float64).reshape(-1), y_var


# --------------------------------------------------------------------------------------
# Row 2: kernel matrix (GPflow Matern32.K as called from models.py:186-189,198,217)
# --------------------------------------------------------------------------------------
def sq_dist(X, ls):
    Xs = X / ls
    s = np.sum(Xs * Xs, axis=1)
    return -2.0 * (Xs @ Xs.T) + s[:, None] + s[None, :]


def kernel_matrix(X, variance, lengthscale, kernel=KERNEL_MATERN32):
    """K0 (no jitter) and dK/d lengthscale.  r = sqrt(max(r2, 1e-36)) as GPflow does."""
    r2 = sq_dist(X, lengthscale)
    if kernel == KERNEL_RBF:
        r2c = np.maximum(r2, 0.0)
        K = variance * np.exp(-0.5 * r2c)
        dK_dl = K * r2c / lengthscale
        return K, dK_dl
    clamp = r2 <= 1e-36
    r = np.sqrt(np.maximum(r2, 1e-36))
    if kernel == KERNEL_MATERN32:
        e = np.exp(-SQRT3 * r)
        K = variance * (1.0 + SQRT3 * r) * e
        dK_dl = 3.0 * variance * r * r * e / lengthscale
    elif kernel == KERNEL_MATERN52:
        e = np.exp(-SQRT5 * r)
        K = variance * (1.0 + SQRT5 * r + (5.0 / 3.0) * r * r) * e
        dK_dl = (5.0 / 3.0) * variance * r * r * (1.0 + SQRT5 * r) * e / lengthscale
    else:
        raise ValueError("unknown kernel id %r" % (kernel,))
    dK_dl = np.where(clamp, 0.0, dK_dl)
    return K, dK_dl


# --------------------------------------------------------------------------------------
# Rows 3-6: the fit (models.py:185-220)
# --------------------------------------------------------------------------------------
class FitState:
    """Variational + optimiser state of one member."""

    def __init__(self, N):
        self.Lam = np.eye(N)              # precision of q(v): q_sqrt = I  (VGP init)
        self.theta1 = np.zeros(N)         # S^-1 m : q_mu = 0
        self.u = np.array([SOFTPLUS_INV_ONE, SOFTPLUS_INV_ONE])   # (u_var, u_ls)
        self.adam_m = np.zeros(2)
        self.adam_v = np.zeros(2)
        self.m = np.zeros(N)
        self.C = np.eye(N)
        self.t = 0


def natgrad_step(st, L, y, nu, gamma=0.5):
    """Closed form of NaturalGradient(gamma).minimize on the whitened VGP   [models.py:191,209]."""
    W = L / np.sqrt(nu)[:, None]
    G = W.T @ W
    N = L.shape[0]
    st.Lam = (1.0 - gamma) * st.Lam + gamma * (np.eye(N) + G)
    st.theta1 = (1.0 - gamma) * st.theta1 + gamma * (L.T @ (y / nu))
    st.C = np.linalg.cholesky(st.Lam)
    z = sla.solve_triangular(st.C, st.theta1, lower=True)
    st.m = sla.solve_triangular(st.C, z, lower=True, trans="T")


def hyper_gradient(st, X, K0, dK_dl, L, y, nu, variance):
    """d(-ELBO)/d(variance, lengthscale) with (m, S) held fixed       [models.py:142-149,210]."""
    m = st.m
    # X_ = L S = L C^-T C^-1
    Y = sla.solve_triangular(st.C, L.T, lower=True).T            # L C^-T
    LS = sla.solve_triangular(st.C, Y.T, lower=True, trans="T").T  # (L C^-T) C^-1
    a = L @ m - y
    Lbar = np.tril((np.outer(a, m) + LS) / nu[:, None])
    P = L.T @ Lbar
    Phi = np.tril(P)
    Phi[np.diag_indices_from(Phi)] *= 0.5
    Wm = sla.solve_triangular(L, Phi.T, lower=True, trans="T").T   # Phi L^-1
    Kbar = sla.solve_triangular(L, Wm, lower=True, trans="T")      # L^-T Phi L^-1
    g_var = np.sum(Kbar * K0) / variance
    g_ls = np.sum(Kbar * dK_dl)
    return np.array([g_var, g_ls])


def adam_step(st, g_u, lr=0.01, b1=0.9, b2=0.999, eps=1e-7):
    """TF 2.8.1 Keras Adam (non-amsgrad)                              [models.py:192,210]."""
    st.t += 1
    st.adam_m = b1 * st.adam_m + (1.0 - b1) * g_u
    st.adam_v = b2 * st.adam_v + (1.0 - b2) * g_u * g_u
    lr_t = lr * math.sqrt(1.0 - b2 ** st.t) / (1.0 - b1 ** st.t)
    st.u = st.u - lr_t * st.adam_m / (np.sqrt(st.adam_v) + eps)


def neg_elbo(st, L, y, nu):
    """-ELBO of the whitened VGP with the heteroskedastic Gaussian likelihood [models.py:142-149,211-215]."""
    N = L.shape[0]
    Lm = L @ st.m
    Cinv_Lt = sla.solve_triangular(st.C, L.T, lower=True)         # C^-1 L^T ; (L S L^T)_tt = colnorm^2
    fvar = np.sum(Cinv_Lt * Cinv_Lt, axis=0)
    data = np.sum(0.5 * LOG2PI + 0.5 * np.log(nu) + 0.5 * ((y - Lm) ** 2 + fvar) / nu)
    logdetS = -2.0 * np.sum(np.log(np.diag(st.C)))
    Cinv = sla.solve_triangular(st.C, np.eye(N), lower=True)
    trS = np.sum(Cinv * Cinv)
    kl = 0.5 * (st.m @ st.m - N - logdetS + trS)
    return data + kl


def gp_fit(X, y_mean, y_var, n_iters, kernel=KERNEL_MATERN32, gamma=0.5, lr=0.01,
           jitter=JITTER, return_state=False, loss_every=0):
    """GPDTW1D.fit after the DBA stage: (X, y_mean, y_var) -> (mu, cov, hypers) [models.py:185-220].

    One iteration = natgrad step at the current hypers, then one Adam step on (u_var, u_ls)
    with (m, S) fixed (models.py:208-210).  Prediction uses the final hypers and the (m, S) of the
    last natgrad step (models.py:217), then adds diag(y_var) (models.py:220).
    """
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y_mean, dtype=np.float64).reshape(-1)
    nu = np.asarray(y_var, dtype=np.float64).reshape(-1)
    N = X.shape[0]
    st = FitState(N)
    losses = []
    for it in range(n_iters):
        variance, ls = softplus(st.u[0]), softplus(st.u[1])
        K0, dK_dl = kernel_matrix(X, variance, ls, kernel)
        L = np.linalg.cholesky(K0 + jitter * np.eye(N))
        natgrad_step(st, L, y, nu, gamma)
        g = hyper_gradient(st, X, K0, dK_dl, L, y, nu, variance)
        g_u = g * (1.0 - np.exp(-np.array([variance, ls])))     # softplus'(u) = 1 - exp(-softplus(u))
        adam_step(st, g_u, lr)
        if loss_every and it % loss_every == 0:
            v2, l2 = softplus(st.u[0]), softplus(st.u[1])
            K2, _ = kernel_matrix(X, v2, l2, kernel)
            L2 = np.linalg.cholesky(K2 + jitter * np.eye(N))
            losses.append(neg_elbo(st, L2, y, nu))
    variance, ls = softplus(st.u[0]), softplus(st.u[1])
    mu, cov = predict_full_cov(st, X, nu, variance, ls, kernel, jitter)
    out = (mu, cov, np.array([variance, ls]))
    if loss_every:
        out = out + (np.array(losses),)
    if return_state:
        out = out + (st,)
    return out


def predict_full_cov(st, X, nu, variance, ls, kernel=KERNEL_MATERN32, jitter=JITTER):
    """predict_f(X, full_cov=True) + diag(y_var)                      [models.py:217-220].

    GPflow conditional(white=True): Kmm = K0 + jitter I = L L^T, A = L^-1 K0 (K0 *without* jitter),
    mu = A^T q_mu, cov = K0 - A^T A + A^T S A.
    """
    N = X.shape[0]
    K0, _ = kernel_matrix(X, variance, ls, kernel)
    L = np.linalg.cholesky(K0 + jitter * np.eye(N))
    A = sla.solve_triangular(L, K0, lower=True)
    mu = A.T @ st.m
    A2 = sla.solve_triangular(st.C, A, lower=True)
    cov = K0 - A.T @ A + A2.T @ A2 + np.diag(nu)
    return mu, cov


# --------------------------------------------------------------------------------------
# Row 7: Distribution.__post_init__ for the full-covariance MVN   (data.py:38-39)
# --------------------------------------------------------------------------------------
def mvn_factor(cov):
    """distrax MultivariateNormalFullCovariance caches chol(cov) and sum(log diag)  [data.py:38-39]."""
    Lc = np.linalg.cholesky(cov)
    return Lc, float(np.sum(np.log(np.diag(Lc))))


# --------------------------------------------------------------------------------------
# Row 8: LogLikelihoodWeight  (weights.py:87-131)
# --------------------------------------------------------------------------------------
def mvn_constvec_loglik(mu, Lc, logdet, obs):
    """ll[r,t] = log N(obs[r,t] * ones(N); mu, cov)                   [weights.py:98-100].

    The reference passes an (N,1) array to an event-shape-N MVN; broadcasting evaluates the density
    at the constant vector x_t * 1 for every t.
    """
    obs = np.atleast_2d(np.asarray(obs, dtype=np.float64))
    N = mu.shape[0]
    a = sla.solve_triangular(Lc, np.ones(N), lower=True)
    c = sla.solve_triangular(Lc, mu, lower=True)
    # (R_obs, N_t, N): z = x * a - c
    z = obs[:, :, None] * a[None, None, :] - c[None, None, :]
    q = np.sum(z * z, axis=-1)
    return -0.5 * q - logdet - 0.5 * N * LOG2PI


def mvn_logprob(mu, Lc, logdet, x):
    """distrax MultivariateNormalFullCovariance.log_prob for ordinary arguments x (..., N):
    -|Lc^-1 (x - mu)|^2 / 2 - sum(log diag Lc) - N log(2 pi) / 2   [distrax 0.1.2; data.py:38-39 builds the object]."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    z = sla.solve_triangular(Lc, (x - mu).T, lower=True)
    return -0.5 * np.sum(z * z, axis=0) - logdet - 0.5 * mu.shape[0] * LOG2PI


def normal_loglik(mu, scale, obs):
    """dx.Normal(loc, scale).log_prob(x) elementwise                  [weights.py:95-96].

    NB the reference feeds a *variance* as ``scale`` (data.py:39)."""
    obs = np.atleast_2d(np.asarray(obs, dtype=np.float64))
    z = (obs - mu[None, :]) / scale[None, :]
    return -0.5 * z * z - np.log(scale)[None, :] - 0.5 * LOG2PI


def loglik_mean_rows(ll):
    """mean over observation realisations                             [weights.py:103-104]."""
    return np.mean(ll, axis=0)


def loglik_weights(ll_mean, c_std=1.0):
    """exp(c * ll_mean) normalised over the model axis                [weights.py:107,118-123].

    ll_mean (M,N).  Returns (weights, unnormalised).  Under/overflow behaviour (0/0 = NaN) is
    deliberately the reference's."""
    with np.errstate(all="ignore"):
        raw = np.exp(c_std * np.asarray(ll_mean, dtype=np.float64))
        w = raw / raw.sum(axis=0, keepdims=True)
    return w, raw


# --------------------------------------------------------------------------------------
# Row 9: ModelSimilarityWeight + gaussian_w2_distance_distrax  (weights.py:240-333, wasserstein.py:10-47)
# --------------------------------------------------------------------------------------
def sqrtm_svd(A):
    """wasserstein.py:10-13."""
    u, s, vt = np.linalg.svd(A)
    return u @ np.diag(np.sqrt(s)) @ vt


def gaussian_w2(mu1, sigma1, mu2, sigma2):
    """wasserstein.py:21-47 (note: location gap is the *un-squared* 2-norm)."""
    location_gap = np.linalg.norm(mu1 - mu2, ord=2)
    s1h = sqrtm_svd(sigma1)
    gap = sigma1 + sigma2 - 2.0 * sqrtm_svd(s1h @ sigma2 @ s1h)
    return location_gap + np.trace(gap)


def w2_weights_single(mus, covs):
    """mode='single'                                                   [weights.py:240-265,331]."""
    M = len(mus)
    D = np.zeros((M, M))
    for i in range(M):
        for j in range(M):
            D[i, j] = gaussian_w2(mus[i], covs[i], mus[j], covs[j])
    v = np.nanmean(D, axis=1)
    return v / v.sum(), D


def w2_weights_temporal(mean, var):
    """mode='temporal'                                                 [weights.py:302-325,331].

    mean, var (M,N) = ._dist.mean(), ._dist.variance() of every member.  Per point the reference
    builds dx.Normal(mean_t, scale=variance_t) whose variance() is variance_t**2, so
    D[i,j,t] = |mu_it - mu_jt| + (v_it - v_jt)**2."""
    mean = np.asarray(mean, dtype=np.float64)
    var = np.asarray(var, dtype=np.float64)
    D = np.abs(mean[:, None, :] - mean[None, :, :]) + (var[:, None, :] - var[None, :, :]) ** 2
    v = np.mean(D, axis=1)
    return v / v.sum(axis=0, keepdims=True)


def w2_weights_spatial(mean, V):
    """mode='spatial', dx.Normal posteriors                             [weights.py:267-300,331; wasserstein.py:21-47].

    mean, V (M, T, n_lat, n_lon) = Distribution.mean / Distribution.variance of every member.  Per (i, j, lat, lon) the
    reference builds dx.Normal(mean_i[:, lat, lon], scale=V_i[:, lat, lon]) and calls gaussian_w2_distance_distrax with
    full_cov=False: sigma = diag(Normal.variance()) = diag(V^2), sqrtm(diag) = diag(V), so
    D = ||mean_i - mean_j||_2 + trace(V_i^2 + V_j^2 - 2 V_i V_j) = ||dmean||_2 + sum_t (V_i - V_j)^2.
    w = nanmean_j D, normalised over the models -> (M, n_lat, n_lon)."""
    mean, V = np.asarray(mean, dtype=np.float64), np.asarray(V, dtype=np.float64)
    M = mean.shape[0]
    D = np.full((M, M) + mean.shape[2:], np.nan)
    for i in range(M):
        for j in range(M):
            D[i, j] = np.sqrt(np.sum((mean[i] - mean[j]) ** 2, axis=0)) + np.sum((V[i] - V[j]) ** 2, axis=0)
    v = np.nanmean(D, axis=1)
    return v / v.sum(axis=0, keepdims=True)


# --------------------------------------------------------------------------------------
# Row 10: Barycentre  (ensemble_scheme.py:43-81, wasserstein.py:61-100)
# --------------------------------------------------------------------------------------
def gaussian_barycentre(means, std_devs, weights, tolerance=1e-6, init_var=1.0):
    """wasserstein.py:61-100 verbatim semantics incl. the *signed* convergence test.

    Returns (mu, sigma, converged)."""
    v = init_var
    n_iters = 0
    converged = True
    while True:
        cand = 0.0
        for w, s in zip(weights, std_devs):
            cand += w * math.sqrt(v) * s
        if cand - v < tolerance:
            v = cand
            break
        v = cand
        n_iters += 1
        if n_iters > 200:
            converged = False
            break
    mu = float(np.sum(np.asarray(weights) * np.asarray(means)))
    return mu, math.sqrt(v), converged


def barycentre(weights, mean, var, tolerance=1e-6, init_var=1.0):
    """ensemble_scheme.py:52-78: per point t barycentre of N(mean[m,t], var[m,t]).

    Returns (mu (P,), covariance = sigma**2 (P,), n_not_converged)."""
    weights = np.asarray(weights, dtype=np.float64)
    M, P = mean.shape
    weights = weights.reshape(M, -1)
    mu = np.empty(P)
    sig = np.empty(P)
    bad = 0
    std = np.sqrt(var)
    for t in range(P):
        mu[t], sig[t], ok = gaussian_barycentre(mean[:, t], std[:, t], weights[:, t], tolerance, init_var)
        bad += 0 if ok else 1
    return mu, sig ** 2, bad


# --------------------------------------------------------------------------------------
# Row 11: WeightedModelMean  (ensemble_scheme.py:136-160)
# --------------------------------------------------------------------------------------
def weighted_model_mean(datas, weights):
    """datas: list of (R_m,P) arrays, weights (M,P)."""
    weights = np.asarray(weights, dtype=np.float64)
    wm = 0.0
    wv = 0.0
    for m, d in enumerate(datas):
        d = np.asarray(d, dtype=np.float64)
        mean = d.mean(axis=0)
        var = d.std(axis=0) ** 2
        wv = wv + var * weights[m] ** 2
        wm = wm + mean * weights[m]
    return wm, wv


# --------------------------------------------------------------------------------------
# "Next" rows 8(f)-1: the remaining weight plug-ins and MultiModelMean
# --------------------------------------------------------------------------------------
def crps_weights(mean, var, obs):
    """CRPSWeight (weights.py:444-515): target dx.Normal(mean_i, scale=variance_i), properscoring.crps_gaussian
    with sig = stddev() = variance_i; score = mean over observation realisations; weight ~ 1/score."""
    from scipy.stats import norm
    mean, var, obs = np.asarray(mean), np.asarray(var), np.atleast_2d(obs)
    z = (obs[None, :, :] - mean[:, None, :]) / var[:, None, :]
    crps = var[:, None, :] * (z * (2.0 * norm.cdf(z) - 1.0) + 2.0 * norm.pdf(z) - 1.0 / math.sqrt(math.pi))
    score = crps.mean(axis=1)
    inv = 1.0 / score
    return inv / inv.sum(axis=0, keepdims=True), score


def ksd_weights(mean, var, obs, c=1.0, beta=-0.5):
    """KSDWeight (weights.py:336-441): IMQ-kernel Stein discrepancy of the R_obs observation values at each point
    against dx.Normal(mean_i, scale=variance_i) (so grad log p = -(x - mean)/variance^2); weight ~ 1/KSD."""
    mean, var, obs = np.asarray(mean), np.asarray(var), np.atleast_2d(obs)
    M, P = mean.shape
    R = obs.shape[0]
    ksd = np.empty((M, P))
    for m in range(M):
        for t in range(P):
            x = obs[:, t]
            g = -(x - mean[m, t]) / var[m, t] ** 2
            tot = 0.0
            for a in range(R):
                diff = x[a] - x
                q = c * c + diff * diff
                k0 = (g[a] * g * q ** beta - 2 * beta * g[a] * diff * q ** (beta - 1) + 2 * beta * g * diff * q ** (beta - 1)
                      - 2 * 1 * beta * q ** (beta - 1) - 4 * beta * (beta - 1) * q ** (beta - 2) * diff * diff)
                tot += k0.sum()
            ksd[m, t] = math.sqrt(tot) / R
    inv = 1.0 / ksd
    return inv / inv.sum(axis=0, keepdims=True), ksd


def inverse_square_weights(datas, obs):
    """InverseSquareWeight (weights.py:157-169): (mean_r model - mean_r obs)^-2 normalised over the models."""
    om = np.atleast_2d(obs).mean(axis=0)
    raw = np.stack([(np.asarray(d).mean(axis=0) - om) ** -2.0 for d in datas])
    return raw / raw.sum(axis=0, keepdims=True)


def multi_model_mean(datas):
    """MultiModelMean (ensemble_scheme.py:108-128): mean and std**2 over all realisations of all models."""
    allv = np.concatenate([np.asarray(d) for d in datas], axis=0)
    return allv.mean(axis=0), allv.std(axis=0) ** 2


# --------------------------------------------------------------------------------------
# distrax log_prob of the diagonal ensemble posterior (utils.py:139; Appendix C of SURVEY)
# --------------------------------------------------------------------------------------
def mvn_diag_logprob(loc, scale_diag, x):
    x = np.atleast_2d(x)
    z = (x - loc[None, :]) / scale_diag[None, :]
    return np.sum(-0.5 * z * z - np.log(scale_diag)[None, :] - 0.5 * LOG2PI, axis=1)


# --------------------------------------------------------------------------------------
# Seeded synthetic ensemble; the product has the identical generator on its side
# (bayesian_ensembling_amd/synthetic.py) - the two are compared in tests.
# --------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _M64

    def next_u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def uniform(self):
        return ((self.next_u64() >> 11) + 0.5) * (1.0 / 9007199254740992.0)

    def normal(self):
        u1 = self.uniform()
        u2 = self.uniform()
        return math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2)


def synthetic_ensemble(M, N, R=8, R_obs=3, seed=0xB4E5):
    """Returns (data (M,R,N), obs (R_obs,N)).  tau=t/(N-1); member signal 1.5*a*tau^2+b; AR(1) noise."""
    rng = SplitMix64(seed)
    tau = np.arange(N, dtype=np.float64) / max(N - 1, 1)
    data = np.empty((M, R, N))

    def ar1():
        e = np.empty(N)
        prev = 0.0
        for t in range(N):
            prev = 0.6 * prev + 0.12 * rng.normal()
            e[t] = prev
        return e

    for m in range(M):
        a = 0.8 + 0.4 * rng.uniform()
        b = 0.1 * rng.normal()
        sig = 1.5 * a * tau * tau + b
        for r in range(R):
            data[m, r] = sig + ar1()
    obs = np.empty((R_obs, N))
    for r in range(R_obs):
        obs[r] = 1.5 * tau * tau + ar1()
    return data, obs
