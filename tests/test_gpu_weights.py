"""Weights + weighted posterior: HIP kernels vs the oracle, reference quirks included, and the plug-in API
exercised the way the reference's own tests/test_weights.py:71-101 does (type, shape, sum over 'model' == 1)."""
import numpy as np
import pytest

import bayesian_ensembling_amd as es
from bayesian_ensembling_amd import distributions as dx
from bayesian_ensembling_amd.labelled import DataArray
from conftest import member_inputs, relerr
from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu


def _posteriors(ctx, M, N, R=4, seed=2, T=2):
    data, obs = go.synthetic_ensemble(M, N, R=R, R_obs=3, seed=seed)
    X, y, nu = member_inputs(data)
    return data, obs, ctx.gp_fit_batch(X, y, nu, T)


@pytest.mark.parametrize("M,N,R_obs", [(8, 200, 3), (2, 86, 1), (3, 130, 10), (2, 7, 5)])
def test_loglik_rows_and_weights(ctx, M, N, R_obs):
    data, _, out = _posteriors(ctx, M, N)
    _, obs = go.synthetic_ensemble(1, N, R=2, R_obs=R_obs, seed=9)
    ll = ctx.loglik_rows(out["mu"], out["chol"], out["logdet"], obs)
    ll_ref = np.stack([go.loglik_mean_rows(go.mvn_constvec_loglik(out["mu"][m], out["chol"][m], out["logdet"][m], obs))
                       for m in range(M)])
    assert relerr(ll, ll_ref) < 1e-12
    # the basis path (bordered Cholesky by-product, expanded quadratic form) must agree with the direct form
    assert relerr(ctx.loglik_rows_basis(out["basis"], out["logdet"], obs), ll_ref) < 1e-11
    chol2, ld2, basis2 = ctx.mvn_factor_batch(out["cov"], out["mu"])
    assert relerr(basis2, out["basis"]) < 1e-11 and relerr(chol2, out["chol"]) < 1e-12
    for c in (1e-3, 1e-2):
        w, raw = ctx.weights_normalise(ll, c, return_raw=True)
        w_ref, raw_ref = go.loglik_weights(ll_ref, c)
        assert relerr(w, w_ref) < 1e-12 and relerr(raw, raw_ref) < 1e-12
        assert np.allclose(w.sum(axis=0), 1.0, rtol=1e-12)


@pytest.mark.parametrize("M,N", [(3, 1000), (2, 64), (5, 191), (40, 130), (2500, 86), (1, 1)])
def test_loglik_rows_chain_reads_only_the_lower_triangle(ctx, M, N):
    """bayens_loglik_rows walks the compact factor in ONE chained launch (one workgroup per 64 rows and member): whatever
    sits above the diagonal must not matter, partial last blocks and more members than a wave of workgroups included."""
    from scipy.linalg import solve_triangular
    rng = np.random.default_rng(N)
    A = rng.standard_normal((M, N, N + 8))
    cov = A @ A.transpose(0, 2, 1) / N + 0.5 * np.eye(N)
    chol = np.linalg.cholesky(cov)
    logdet = 2.0 * np.log(np.diagonal(chol, axis1=1, axis2=2)).sum(axis=1)
    mu = rng.standard_normal((M, N))
    obs = rng.standard_normal((4, N))
    ll_ref = np.stack([go.loglik_mean_rows(go.mvn_constvec_loglik(mu[m], chol[m], logdet[m], obs)) for m in range(M)])
    dirty = chol + np.triu(np.full((N, N), np.nan), 1)
    ll = ctx.loglik_rows(mu, dirty, logdet, obs)
    assert relerr(ll, ll_ref) < 1e-11
    z = np.stack([solve_triangular(chol[m], np.stack([np.ones(N), mu[m]], axis=1), lower=True).T for m in range(M)])
    assert relerr(ctx.loglik_rows_basis(z, logdet, obs), ll_ref) < 1e-11


def test_weighted_posterior_in_one_call_equals_the_separate_steps(ctx):
    data, obs, out = _posteriors(ctx, 5, 90, R=4, seed=17)
    w, bm, bv, bad = ctx.weighted_posterior(out["basis"], out["logdet"], obs, out["mu"], out["cov"], c_std=1e-3)
    ll = ctx.loglik_rows_basis(out["basis"], out["logdet"], obs)
    w_ref = ctx.weights_normalise(ll, 1e-3)
    bm_ref, bv_ref, bad_ref = ctx.barycentre(w_ref, out["mu"], ctx.extract_diag(out["cov"]))
    assert np.array_equal(w, w_ref) and np.array_equal(bm, bm_ref) and np.array_equal(bv, bv_ref) and bad == bad_ref


def test_exp_underflow_gives_nan_like_the_reference(ctx):
    ll = np.array([[-4000.0, -10.0], [-3000.0, -12.0]])
    w = ctx.weights_normalise(ll, 1.0)
    w_ref, _ = go.loglik_weights(ll, 1.0)
    assert np.isnan(w[:, 0]).all() and np.array_equal(np.isnan(w), np.isnan(w_ref))
    assert relerr(w[:, 1], w_ref[:, 1]) < 1e-14


def test_normal_branch(ctx):
    rng = np.random.default_rng(0)
    mu, scale, obs = rng.standard_normal((3, 50)), rng.uniform(0.1, 2, (3, 50)), rng.standard_normal((4, 50))
    ll = ctx.normal_loglik_rows(mu, scale, obs)
    ref = np.stack([go.normal_loglik(mu[m], scale[m], obs).mean(axis=0) for m in range(3)])
    assert relerr(ll, ref) < 1e-13


def test_w2_temporal(ctx):
    rng = np.random.default_rng(1)
    mean, var = rng.standard_normal((5, 333)), rng.uniform(0.01, 3, (5, 333))
    mean[1] = mean[0]                       # a tie: identical marginals, zero distance between the pair
    var[1] = var[0]
    assert relerr(ctx.w2_temporal(mean, var), go.w2_weights_temporal(mean, var)) < 1e-13


@pytest.mark.parametrize("M,N", [(4, 40), (3, 100), (5, 64), (4, 1000)])    # (4, 1000): fitted covariances at size - the
# polar iteration's convergence depends on their conditioning (6 pairs; the oracle's SVD takes ~1 s per pair)
def test_w2_single_full_covariance(ctx, M, N):
    """mode='single' (weights.py:240-265): all-pairs Gaussian W2 with SVD-based sqrtm in the oracle, polar iteration
    on L_j^T L_i on the device."""
    data, _, out = _posteriors(ctx, M, N, R=5, seed=11)
    w, D = ctx.w2_single(out["mu"], chol=out["chol"])
    w_ref, D_ref = go.w2_weights_single(list(out["mu"]), list(out["cov"]))
    off = ~np.eye(M, dtype=bool)
    assert relerr(D[off], D_ref[off]) < 1e-9
    assert np.all(np.abs(D_ref[~off]) < 1e-9) and not D[~off].any()      # i == j: exactly 0 here, ~1e-12 noise there
    assert relerr(w, w_ref) < 1e-9 and abs(w.sum() - 1.0) < 1e-14
    assert np.allclose(D, D.T, rtol=0, atol=0)


def _nuclear_reference(Li, Lj):
    return np.linalg.svd(Lj.T @ Li, compute_uv=False).sum()


@pytest.mark.parametrize("case", ["scaled_down", "scaled_up", "ill_conditioned", "mixed_scales", "identical", "one_pair",
                                  "more_pairs_than_slots"])
def test_w2_single_starting_interval_and_stopping_rule(ctx, case):
    """The polar iteration starts from per-member estimates of the factors' extreme singular values and stops on the size
    of its steps: badly scaled, ill-conditioned, coincident and very unequal members must neither fool the estimates nor
    end the iteration early.  Reference: nuclear norms from an SVD (wasserstein.py:41-45 is tr sqrtm = ||L_j^T L_i||_*)."""
    rng = np.random.default_rng(sum(case.encode()))
    N = 96
    M = {"one_pair": 2, "more_pairs_than_slots": 9}.get(case, 4)

    def factor(scale=1.0, cond=1e2):
        Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
        ev = scale * np.logspace(0, -np.log10(cond), N)
        return np.linalg.cholesky((Q * ev) @ Q.T + 0.0)
    if case == "scaled_down":
        Ls = [factor(1e-8) for _ in range(M)]
    elif case == "scaled_up":
        Ls = [factor(1e6) for _ in range(M)]
    elif case == "ill_conditioned":
        Ls = [factor(1.0, 1e9) for _ in range(M)]
    elif case == "mixed_scales":
        Ls = [factor(10.0 ** (3 * m - 4), 10.0 ** (1 + 2 * m)) for m in range(M)]
    elif case == "identical":
        Ls = [factor()] * 2 + [factor(), factor(3.0)]
    else:
        Ls = [factor(1.0 + m, 10.0 ** (1 + m % 4)) for m in range(M)]
    chol = np.stack(Ls)
    mu = rng.standard_normal((M, N))
    w, D = ctx.w2_single(mu, chol=chol)
    tr = np.array([(L * L).sum() for L in Ls])
    for i in range(M):
        for j in range(M):
            if i == j:
                assert D[i, j] == 0.0
                continue
            nuc = _nuclear_reference(Ls[i], Ls[j])
            ref = np.linalg.norm(mu[i] - mu[j]) + (tr[i] + tr[j] - 2.0 * nuc)
            # the trace term is a difference of numbers of size tr_i + tr_j: that is the scale of its rounding error
            assert abs(D[i, j] - ref) <= 1e-11 * (tr[i] + tr[j]) + 1e-12 * abs(ref), (case, i, j, D[i, j], ref)
    v = D.mean(axis=1)
    assert relerr(w, v / v.sum()) < 1e-12


def test_w2_single_sharded_over_a_one_rank_communicator(ctx):
    """The multi-GPU form of mode='single' (all-gather of the factors, pairs dealt over ranks, all-reduce of the
    nuclear-norm table) on a communicator of one rank: must equal the single-GPU table exactly."""
    from bayesian_ensembling_amd import parallel
    data, _, out = _posteriors(ctx, 4, 50, R=5, seed=13)
    w_ref, D_ref = ctx.w2_single(out["mu"], chol=out["chol"])
    ctx.comm_init(0, 1, ctx.comm_unique_id())
    try:
        w, D = parallel.w2_single_sharded(ctx, 4, out["mu"], out["chol"])
    finally:
        ctx.comm_destroy()
    assert np.array_equal(D, D_ref) and np.array_equal(w, w_ref)


def test_w2_single_diagonal_posteriors(ctx):
    rng = np.random.default_rng(12)
    M, N = 4, 33
    mu, sd = rng.standard_normal((M, N)), rng.uniform(0.1, 2.0, (M, N))
    w, D = ctx.w2_single(mu, sd=sd)
    D_ref = np.array([[go.gaussian_w2(mu[i], np.diag(sd[i] ** 2), mu[j], np.diag(sd[j] ** 2)) for j in range(M)]
                      for i in range(M)])
    assert relerr(D, D_ref) < 1e-10
    v = D_ref.mean(axis=1)
    assert relerr(w, v / v.sum()) < 1e-10


def test_crps_ksd_inverse_square_and_multi_model_mean(ctx):
    rng = np.random.default_rng(21)
    M, P, R_obs = 4, 57, 5
    mean, var, obs = rng.standard_normal((M, P)), rng.uniform(0.05, 1.5, (M, P)), rng.standard_normal((R_obs, P))
    w, sc = ctx.score_weights("crps", mean, var, obs)
    w_ref, sc_ref = go.crps_weights(mean, var, obs)
    assert relerr(sc, sc_ref) < 1e-12 and relerr(w, w_ref) < 1e-12
    w, sc = ctx.score_weights("ksd", mean, var, obs)
    w_ref, sc_ref = go.ksd_weights(mean, var, obs)
    assert relerr(sc, sc_ref) < 1e-11 and relerr(w, w_ref) < 1e-11
    R = np.array([3, 5, 2, 4], dtype=np.int32)
    datas = [rng.standard_normal((r, P)) for r in R]
    packed = np.zeros((M, 5, P))
    for i, d in enumerate(datas):
        packed[i, :R[i]] = d
    assert relerr(ctx.inverse_square_weights(packed, R, obs), go.inverse_square_weights(datas, obs)) < 1e-12
    mu, v = ctx.multi_model_mean(packed, R)
    rm, rv = go.multi_model_mean(datas)
    assert relerr(mu, rm) < 1e-13 and relerr(v, rv) < 1e-13


def test_barycentre_both_sides_of_the_signed_test(ctx):
    rng = np.random.default_rng(3)
    M, P = 6, 500
    w = rng.random((M, P))
    w /= w.sum(axis=0)
    mean = rng.standard_normal((M, P))
    var = rng.uniform(0.01, 9.0, (M, P))          # sum w sigma straddles 1: both branches of wasserstein.py:85-91
    mu, v, bad = ctx.barycentre(w, mean, var)
    rm, rv, rbad = go.barycentre(w, mean, var)
    S = (w * np.sqrt(var)).sum(axis=0)
    assert (S < 1).any() and (S > 1).any() and bad == rbad == 0
    assert relerr(mu, rm) < 1e-14 and relerr(v, rv) < 1e-13
    # non-finite weights (the reference's exp underflow) never converge: counted, not hidden
    w[:, 0] = np.nan
    _, _, bad = ctx.barycentre(w, mean, var)
    assert bad == 1


def test_weighted_model_mean_with_ragged_realisations(ctx):
    rng = np.random.default_rng(4)
    R = np.array([3, 5, 2], dtype=np.int32)
    P = 40
    datas = [rng.standard_normal((r, P)) for r in R]
    w = rng.random((3, P))
    packed = np.zeros((3, 5, P))
    for i, d in enumerate(datas):
        packed[i, :R[i]] = d
    mu, v = ctx.weighted_model_mean(packed, R, w)
    rm, rv = go.weighted_model_mean(datas, w)
    assert relerr(mu, rm) < 1e-14 and relerr(v, rv) < 1e-14


# ---- plug-in API, as tests/test_weights.py of the reference drives it ---------------------------------
def _collection(n_models, obs_n_reals, N=24, R=3, seed=0):
    rng = np.random.default_rng(seed)
    time = np.arange("1960-01", "1962-01", dtype="datetime64[M]")[:N]

    def da(n_reals):
        return DataArray(rng.random((n_reals, N)), ("realisation", "time"),
                         {"realisation": np.arange(n_reals), "time": time})
    mc = es.ModelCollection([es.ProcessModel(da(R), f"model{i}") for i in range(n_models)])
    return mc, es.ProcessModel(da(obs_n_reals), "obs")


def test_reference_data_file_to_posterior(ctx):
    """File -> ProcessModel -> annual anomaly -> GPDTW1D posterior, from one of the reference's own data files
    (N=86 annual steps, 6 realisations), against the oracle."""
    import os
    from conftest import GOLDEN
    da = es.open_dataarray(os.path.join(GOLDEN, "nc_reference_IPSL_ssp126_gmst.nc"))
    clim = np.stack([da.values[:, m::12].mean(axis=1) for m in range(12)]).mean(axis=1)
    anom = es.ProcessModel(da, "IPSL-CM6A-LR").calculate_anomaly(climatology=clim, resample_freq="Y")
    mc = es.ModelCollection([anom])
    mc.fit(model=es.GPDTW1D(), n_optim_nits=30, progress_bar=False)
    X, y, nu = go.prologue(np.asarray(anom.model_data.values))
    mo, co, _ = go.gp_fit(X, y, nu, 30)
    dist = mc.models[0].distribution
    assert relerr(dist.mean.values, mo) < 1e-8 and relerr(np.asarray(dist.covariance), co) < 1e-8


def test_config3_fp32_matern52_with_wasserstein_weights(ctx):
    """BASELINE config 3 through the plug-in API: Matern-5/2 posteriors fitted in fp32, ModelSimilarityWeight
    (temporal and single) and the log-likelihood weights + barycentre, against the float64 oracle: 1e-3 relative."""
    M, N, R = 4, 120, 6
    data, obs = go.synthetic_ensemble(M, N, R=R, R_obs=3, seed=9)
    time = np.arange(N)

    def pm(arr, name):
        return es.ProcessModel(DataArray(arr, ("realisation", "time"), {"realisation": np.arange(arr.shape[0]), "time": time}), name)
    mc = es.ModelCollection([pm(data[m], f"m{m}") for m in range(M)])
    mc.fit(model=es.GPDTW1D(kernel="matern52", dtype="float32"), n_optim_nits=2, progress_bar=False)
    mus, covs = [], []
    for m in range(M):
        X, y, nu = go.prologue(data[m])
        mo, co, _ = go.gp_fit(X, y, nu, 2, kernel=1)
        mus.append(mo); covs.append(co)
        assert relerr(mc.models[m].distribution.mean.values, mo) < 1e-3
    var = np.stack([np.diag(c) for c in covs])
    w_t = es.ModelSimilarityWeight()(mc, pm(obs, "obs"), mode="temporal")
    assert relerr(w_t.values, go.w2_weights_temporal(np.stack(mus), var)) < 1e-3
    w_s = es.ModelSimilarityWeight()(mc, pm(obs, "obs"), mode="single")
    assert relerr(np.asarray(w_s.values).reshape(-1), go.w2_weights_single(mus, covs)[0]) < 1e-3
    w_l = es.LogLikelihoodWeight()(mc, pm(obs, "obs"), standardisation_constant=1e-3)
    ll = np.stack([go.loglik_mean_rows(go.mvn_constvec_loglik(mus[m], *go.mvn_factor(covs[m]), obs)) for m in range(M)])
    assert relerr(w_l.values, go.loglik_weights(ll, 1e-3)[0]) < 1e-3
    post = es.Barycentre()(mc, w_l)
    bm, bv, _ = go.barycentre(go.loglik_weights(ll, 1e-3)[0], np.stack(mus), var)
    assert relerr(post.mean.values, bm) < 1e-3


@pytest.mark.parametrize("obs_n_reals", [1, 2, 5, 10])
@pytest.mark.parametrize("n_models", [2, 5, 10])
def test_weightings_api(ctx, n_models, obs_n_reals):
    mc, obs = _collection(n_models, obs_n_reals)
    mc.fit(model=es.GPDTW1D(), compile_objective=True, n_optim_nits=2, progress_bar=False)
    assert all(m.distribution.losses is None for m in mc.models)       # no progress display, no loss samples
    if n_models == 2 and obs_n_reals == 1:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                            # refitting warns, as in the reference
            mc.fit(model=es.GPDTW1D(), n_optim_nits=27)                # progress_bar defaults to True (models.py:169)
        assert all(m.distribution.losses.shape == (2,) and np.isfinite(m.distribution.losses).all() for m in mc.models)
        mc.fit(model=es.GPDTW1D(), compile_objective=True, n_optim_nits=2, progress_bar=False)
    for weighting in [es.LogLikelihoodWeight(), es.ModelSimilarityWeight()]:
        if weighting.name == "LogLikelihoodWeight":
            weights = weighting(mc, obs, standardisation_constant=0.001)
        else:
            weights = weighting(mc, obs, mode="temporal")
        assert isinstance(weights, DataArray)
        assert weights.shape == (n_models,) + obs.model_data.mean("realisation").shape
        assert np.all(weights.sum("model").values == pytest.approx(1.0, 1e-6))
    # the other four plug-ins of the reference's test (tests/test_weights.py:79-86): type, shape, sum == 1
    for weighting in [es.InverseSquareWeight(), es.UniformWeight(), es.KSDWeight(), es.CRPSWeight()]:
        weights = weighting(mc, obs)
        assert isinstance(weights, DataArray)
        assert weights.shape == (n_models,) + obs.model_data.mean("realisation").shape
        assert np.all(weights.sum("model").values == pytest.approx(1.0, 1e-6))
    mmm = es.MultiModelMean()(mc, None)
    rm, rv = go.multi_model_mean([m.model_data.values for m in mc.models])
    assert relerr(mmm.mu, rm) < 1e-12 and relerr(mmm.covariance, rv) < 1e-12
    mfa = es.MeanFieldApproximation().fit(mc[0])
    assert mfa.dist_type is dx.Normal and relerr(mfa.mu, mc[0].model_data.values.mean(axis=0)) < 1e-13
    assert relerr(mfa.covariance, mc[0].model_data.values.var(axis=0)) < 1e-12
    single = es.ModelSimilarityWeight()(mc)                      # the reference's default mode
    assert single.shape == (n_models, 1) and single.sum("model").values == pytest.approx(1.0, 1e-12)
    w_ref, _ = go.w2_weights_single([m.distribution.mu for m in mc.models], [m.distribution.covariance for m in mc.models])
    assert relerr(single.values[:, 0], w_ref) < 1e-8
    # the whole path against the oracle
    data = np.stack([m.model_data.values for m in mc.models])
    X, y, nu = member_inputs(data)
    w = es.LogLikelihoodWeight()(mc, obs, standardisation_constant=0.001)
    ll_ref = []
    for m in range(n_models):
        mo, co, _ = go.gp_fit(X[m], y[m], nu[m], 2)
        Lc, ld = go.mvn_factor(co)
        ll_ref.append(go.loglik_mean_rows(go.mvn_constvec_loglik(mo, Lc, ld, obs.model_data.values)))
        assert relerr(mc[m].distribution.mean.values, mo) < 1e-9
    w_ref, _ = go.loglik_weights(np.stack(ll_ref), 0.001)
    assert relerr(w.values, w_ref) < 1e-8
    bary = es.Barycentre()(mc, w)
    var = np.stack([np.diag(m.distribution.covariance) for m in mc.models])
    mean = np.stack([m.distribution.mu for m in mc.models])
    rm, rv, _ = go.barycentre(w_ref, mean, var)
    assert bary.dist_type is dx.MultivariateNormalDiag
    assert relerr(bary.mu, rm) < 1e-8 and relerr(bary.covariance, rv) < 1e-8
    assert relerr(bary._dist.variance(), rv ** 2) < 1e-8      # scale_diag = sigma*^2 quirk (ensemble_scheme.py:75-78)
    wmm = es.WeightedModelMean()(mc, w)
    rm2, rv2 = go.weighted_model_mean([m.model_data.values for m in mc.models], w_ref)
    assert relerr(wmm.mu, rm2) < 1e-8 and relerr(wmm.covariance, rv2) < 1e-8


def test_weights_require_fitted_models_and_matching_time(ctx):
    mc, obs = _collection(2, 2)
    with pytest.raises(AssertionError):
        es.LogLikelihoodWeight()(mc, obs)                    # "Distribution not defined - fit models first"
    mc.fit(es.GPDTW1D(), n_optim_nits=1)
    with pytest.warns(UserWarning):                           # refitting warns (data.py:392-393)
        mc.fit(es.GPDTW1D(), n_optim_nits=1)
    obs.model_data.coords["time"] = obs.model_data.coords["time"] + np.timedelta64(31, "D")
    with pytest.raises(AssertionError):
        es.LogLikelihoodWeight()(mc, obs)
    with pytest.raises(NotImplementedError):
        es.GPDTW1D().fit(es.ProcessModel(DataArray(np.random.rand(2, 4, 3), ("realisation", "time", "lat")), "3d"))


def test_return_lls_and_members_with_different_realisation_counts(ctx):
    rng = np.random.default_rng(8)
    time = np.arange(30)
    def da(r): return DataArray(rng.random((r, 30)), ("realisation", "time"), {"realisation": np.arange(r), "time": time})
    mc = es.ModelCollection([es.ProcessModel(da(3), "a"), es.ProcessModel(da(6), "b"), es.ProcessModel(da(3), "c")])
    obs = es.ProcessModel(da(2), "obs")
    mc.fit(es.GPDTW1D(), n_optim_nits=2)
    w, lls = es.LogLikelihoodWeight()(mc, obs, return_lls=True, standardisation_constant=1e-3)
    assert np.allclose((lls / lls.sum("model")).values, w.values)
    assert list(w.coords["model"]) == ["a", "b", "c"]
    one = es.GPDTW1D().fit(mc[1], n_optim_nits=2)
    assert np.array_equal(one.covariance, mc[1].distribution.covariance)


def test_perfect_model_test_metrics_against_the_oracle(ctx, tmp_path):
    """ensembles/utils.py:83-225 (leave-one-out harness): the three metrics of the barycentre and of the MMM."""
    from bayesian_ensembling_amd.utils import PerfectModelTest
    rng = np.random.default_rng(5)
    N, T = 30, 2
    time = np.arange(N)

    def coll(seed):
        r = np.random.default_rng(seed)
        return es.ModelCollection([es.ProcessModel(DataArray(r.random((3, N)) + 0.1 * i, ("realisation", "time"),
                                   {"realisation": np.arange(3), "time": time}), f"m{i}") for i in range(4)])
    hind, fore = coll(1), coll(2)
    pmt = PerfectModelTest(hind, fore, es.GPDTW1D, es.LogLikelihoodWeight, es.Barycentre, ssp="test", save_dir=str(tmp_path))
    header, rows = pmt.run(n_optim_nits=T)
    assert len(rows) == 4 and len(header) == 7 and (tmp_path / "csvs").exists()
    # oracle for the first held-out member
    i = 0
    hd = [m.model_data.values for k, m in enumerate(hind.models) if k != i]
    fd = [m.model_data.values for k, m in enumerate(fore.models) if k != i]
    obs_past, obs_fut = hind[i].model_data.values, fore[i].model_data.values

    def post(d):
        X, y, nu = go.prologue(d)
        mo, co, _ = go.gp_fit(X, y, nu, T)
        return mo, co
    hp, fp = [post(d) for d in hd], [post(d) for d in fd]
    ll = np.stack([go.loglik_mean_rows(go.mvn_constvec_loglik(m_, *go.mvn_factor(c_), obs_past)) for m_, c_ in hp])
    w, _ = go.loglik_weights(ll, 1.0)
    wm = w.mean(axis=1)
    mean = np.stack([m_ for m_, _ in fp])
    var = np.stack([np.diag(c_) for _, c_ in fp])
    bmu, bvar, _ = go.barycentre(np.repeat(wm[:, None], N, axis=1), mean, var)
    nll = -np.mean(go.mvn_diag_logprob(bmu, bvar, obs_fut))
    rmse = np.mean(np.sqrt(np.mean((bmu[:, None] - obs_fut.T) ** 2, axis=0)))
    om, oc = post(obs_fut)
    w2 = go.gaussian_w2(bmu, np.diag(bvar ** 2), om, oc)
    got = rows[0]
    if np.all(np.isfinite(w)):
        assert abs(got[1] - nll) < 1e-7 * abs(nll) and abs(got[2] - rmse) < 1e-9 and abs(got[3] - w2) < 1e-7 * abs(w2)
    allr = np.vstack(fd)
    mm, mv = allr.mean(axis=0), allr.var(axis=0)
    assert abs(got[4] + np.mean(go.normal_loglik(mm, mv, obs_fut))) < 1e-9
    assert abs(got[5] - np.mean(np.sqrt(np.mean((mm[None] - obs_fut) ** 2, axis=0)))) < 1e-12
    assert abs(got[6] - go.gaussian_w2(mm, np.diag(mv ** 2), om, np.diag(np.diag(oc)))) < 1e-8


# ---- round 3: the members' sums at the benchmarked ensemble sizes, the exchange's index arithmetic, size -----------
@pytest.mark.parametrize("M,P", [(32, 2000), (33, 2000), (128, 517), (32, 148608)])
def test_member_sums_at_benchmarked_ensemble_sizes(ctx, M, P):
    """normalise / barycentre / w2_temporal deal the members over four waves per 64 points (kernels.hip); the unit tests
    above stop at M = 10.  Here: M = 32 (configs 2, 3), 33 (not a multiple of the four waves), 128 (config 4's table after
    the all-gather) and P = 1728 x 86 (config 5's per-GPU point axis) - against the oracle's weights.py:107,118-123,
    302-325 and ensemble_scheme.py:52-78."""
    rng = np.random.default_rng(M * 7 + P)
    ll = -300.0 - 2000.0 * rng.random((M, P))
    w, raw = ctx.weights_normalise(ll, 1e-3, return_raw=True)
    w_ref, raw_ref = go.loglik_weights(ll, 1e-3)
    assert relerr(w, w_ref) < 1e-13 and relerr(raw, raw_ref) < 1e-13 and np.allclose(w.sum(axis=0), 1.0, rtol=1e-13)
    mean, var = rng.standard_normal((M, P)), rng.uniform(0.01, 9.0, (M, P))
    mu, v, bad = ctx.barycentre(w, mean, var)
    rm, rv, rbad = go.barycentre(w_ref, mean, var)
    assert bad == rbad == 0 and relerr(mu, rm) < 1e-13 and relerr(v, rv) < 1e-12
    if P <= 2000:
        assert relerr(ctx.w2_temporal(mean, var), go.w2_weights_temporal(mean, var)) < 1e-13


@pytest.mark.parametrize("M,nranks", [(5, 2), (7, 3), (128, 8), (3, 4), (9, 1)])
def test_member_rows_exchange_pack_and_unpack(ctx, M, nranks):
    """The device side of bayens_allgather_member_rows around the collective: pack_rows_kernel per rank into the gathered
    image [nranks][ceil(M/nranks)][3][P] and unpack_rows_kernel back into three [M][P] tables, uneven shards (and ranks
    with no member at all) included - against the host twin parallel.pack_rows / unpack_rows that the gloo tests drive."""
    from bayesian_ensembling_amd import parallel
    P = 301
    rng = np.random.default_rng(M * 10 + nranks)
    ll, mean, var = rng.standard_normal((3, M, P))
    got = ctx.test_member_rows_exchange(nranks, ll, mean, var)
    rows = parallel.padded_rows(M, nranks)
    image = np.concatenate([parallel.pack_rows(ll[lo:hi], mean[lo:hi], var[lo:hi], rows)
                            for lo, hi in (parallel.shard_bounds(M, r, nranks) for r in range(nranks))], axis=0)
    ref = parallel.unpack_rows(image, M, nranks)
    for g, r, src in zip(got, ref, (ll, mean, var)):
        assert np.array_equal(g, r) and np.array_equal(g, src)


def test_allgather_member_rows_on_a_one_rank_communicator(ctx):
    """The exchange entry point itself (pack -> ncclAllGather -> unpack on the context's stream) over RCCL with one rank:
    the three tables must come back bit-identical; a wrong M_loc is refused."""
    M, P = 6, 777
    rng = np.random.default_rng(6)
    tabs = rng.standard_normal((3, M, P))
    d_in = [ctx.to_device(t) for t in tabs]
    d_out = [ctx.empty((M, P)) for _ in range(3)]
    ctx.comm_init(0, 1, ctx.comm_unique_id())
    try:
        assert ctx.comm_info()[:2] == (0, 1)
        ctx.allgather_member_rows(M, P, M, *d_in, *d_out)
        ctx.sync()
        for t, d in zip(tabs, d_out):
            assert np.array_equal(d.download(), t)
        from bayesian_ensembling_amd._lib import BayensError
        with pytest.raises(BayensError):
            ctx.allgather_member_rows(M, P, M - 1, *d_in, *d_out)
        assert ctx.comm_max(3.25) == 3.25
    finally:
        ctx.comm_destroy()
        for d in d_in + d_out:
            d.free()


def test_w2_single_pair_at_the_benchmarked_order(ctx):
    """mode='single' at N = 2000 (bench.py's config-3 weighting step runs 496 such pairs): two fitted members, the polar
    iteration with this size's starting interval and stopping rule against the oracle's SVD formula
    (wasserstein.py:10-13, 21-47)."""
    data, _, out = _posteriors(ctx, 2, 2000, R=8, seed=0xB4E5)
    w, D = ctx.w2_single(out["mu"], chol=out["chol"])
    ref = go.gaussian_w2(out["mu"][0], out["cov"][0], out["mu"][1], out["cov"][1])
    assert abs(D[0, 1] - ref) < 1e-8 * abs(ref) and D[0, 1] == D[1, 0] and D[0, 0] == 0.0 == D[1, 1]
    assert relerr(w, np.array([0.5, 0.5])) < 1e-14


@pytest.mark.parametrize("cond", [1e2, 1e6, 1e8])
def test_log_prob_with_an_ill_conditioned_factor(ctx, cond):
    """bayens_mvn_log_prob inverts the 64 x 64 diagonal blocks of the CALLER's factor (capi.hip: FactorSolve): by direct
    forward substitution (error ~ cond(L) eps), not by re-factoring L L^T (cond^2) - against the oracle's mvn_logprob."""
    rng = np.random.default_rng(int(np.log10(cond)))
    B, N, n_x = 2, 150, 5
    chol = np.empty((B, N, N))
    for b in range(B):
        Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
        sv = np.logspace(0, -np.log10(cond), N)
        Lq = np.linalg.qr(((Q * sv) @ np.linalg.qr(rng.standard_normal((N, N)))[0]).T)[1].T    # lower, singular values sv
        chol[b] = Lq * np.sign(np.diag(Lq))[None, :]
    logdet = np.log(np.abs(np.diagonal(chol, axis1=1, axis2=2))).sum(axis=1)
    mu = rng.standard_normal((B, N))
    z = rng.standard_normal((B, n_x, N))
    x = mu[:, None, :] + np.einsum("bij,bxj->bxi", chol, z)            # so that the quadratic form is O(N), not 1/sv^2
    got = ctx.mvn_log_prob(mu, chol, logdet, x)
    ref = np.stack([go.mvn_logprob(mu[b], chol[b], logdet[b], x[b]) for b in range(B)])
    assert np.allclose(np.linalg.cond(chol[0]), cond, rtol=1e-3)
    assert relerr(got, ref) < 1e-15 * cond * N + 1e-12, relerr(got, ref)


def test_log_prob_refuses_a_factor_with_a_non_positive_diagonal(ctx):
    from bayesian_ensembling_amd._lib import BayensError
    N = 70
    chol = np.stack([np.eye(N), np.eye(N)])
    chol[1, 66, 66] = 0.0
    with pytest.raises(BayensError) as e:
        ctx.mvn_log_prob(np.zeros((2, N)), chol, np.zeros(2), np.zeros((1, N)))
    assert e.value.code == -2 and "member 1" in str(e.value) and "row 67" in str(e.value)
    chol[1, 66, 66] = np.nan
    with pytest.raises(BayensError):
        ctx.mvn_log_prob(np.zeros((2, N)), chol, np.zeros(2), np.zeros((1, N)))


def test_loglik_rows_falls_back_to_the_blocked_solve_when_the_chain_stalls(ctx, tmp_path):
    """bayens_loglik_rows' one-launch substitution chain waits across workgroups with a bounded number of polls; on a
    busy GPU a consumer may give up for valid input.  The entry then recomputes the basis through the blocked solve
    (no cross-workgroup waits) instead of failing.  BAYENS_FORCE_CHAIN_STALL=1 takes that branch deterministically (in
    a child process: the switch is read once per process)."""
    import os
    import subprocess
    import sys
    rng = np.random.default_rng(77)
    M, N = 3, 200
    A = rng.standard_normal((M, N, N + 8))
    chol = np.linalg.cholesky(A @ A.transpose(0, 2, 1) / N + 0.5 * np.eye(N))
    logdet = np.log(np.diagonal(chol, axis1=1, axis2=2)).sum(axis=1)
    mu, obs = rng.standard_normal((M, N)), rng.standard_normal((4, N))
    before = ctx.chain_stalls()
    chained = ctx.loglik_rows(mu, chol, logdet, obs)
    assert ctx.chain_stalls() == before                                          # the chained launch itself went through
    np.savez(tmp_path / "in.npz", mu=mu, chol=chol, logdet=logdet, obs=obs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from bayesian_ensembling_amd import _lib; "
            "d = np.load(%r); c = _lib.Context(0); "
            "np.save(%r, c.loglik_rows(d['mu'], d['chol'], d['logdet'], d['obs'])); "
            "assert c.chain_stalls() == 1"                                      # the fallback is counted, not silent
            % (root, str(tmp_path / "in.npz"), str(tmp_path / "out.npy")))
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, BAYENS_FORCE_CHAIN_STALL="1"), timeout=300)
    fallback = np.load(tmp_path / "out.npy")
    ref = np.stack([go.loglik_mean_rows(go.mvn_constvec_loglik(mu[m], chol[m], logdet[m], obs)) for m in range(M)])
    assert relerr(fallback, ref) < 1e-11 and relerr(chained, ref) < 1e-11
