"""Posteriors that stay in HBM behind the plug-in API (bayesian_ensembling_amd/resident.py): ModelCollection.fit ->
LogLikelihoodWeight -> Barycentre without the covariance ever crossing PCIe; host access, pickling and the general
log_prob still give the reference's values (ensembles/data.py:38-39,385-399; weights.py:87-131; ensemble_scheme.py:43-81)."""
import pickle

import numpy as np
import pytest

import bayesian_ensembling_amd as es
from bayesian_ensembling_amd import distributions as dx
from bayesian_ensembling_amd.labelled import DataArray
from conftest import member_inputs, relerr
from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu


def _collection(reals, N=40, obs_reals=3, seed=3):
    rng = np.random.default_rng(seed)
    time = np.arange("1960-01", "1980-01", dtype="datetime64[M]")[:N]
    tau = np.linspace(0, 1, N)

    def da(n):
        return DataArray(1.5 * tau ** 2 + 0.1 * rng.standard_normal((n, N)).cumsum(axis=1), ("realisation", "time"),
                         {"realisation": np.arange(n), "time": time})
    mc = es.ModelCollection([es.ProcessModel(da(r), f"model{i}") for i, r in enumerate(reals)])
    return mc, es.ProcessModel(da(obs_reals), "obs")


def _oracle_path(mc, obs, T, c_std):
    mus, covs, lls = [], [], []
    for m in mc.models:
        X, y, nu = go.prologue(m.model_data.values)
        mo, co, _ = go.gp_fit(X, y, nu, T)
        Lc, ld = go.mvn_factor(co)
        mus.append(mo); covs.append(co)
        lls.append(go.loglik_mean_rows(go.mvn_constvec_loglik(mo, Lc, ld, obs.model_data.values)))
    w, _ = go.loglik_weights(np.stack(lls), c_std)
    bm, bv, _ = go.barycentre(w, np.stack(mus), np.stack([np.diag(c) for c in covs]))
    return mus, covs, w, bm, bv


@pytest.mark.parametrize("reals", [(4, 4, 4, 4, 4), (3, 5, 3, 8, 5, 3)])     # one batch / ragged R = three batches
def test_weighted_posterior_runs_on_resident_posteriors(ctx, reals):
    mc, obs = _collection(reals)
    mc.fit(model=es.GPDTW1D(), n_optim_nits=2, progress_bar=False)
    dists = [m.distribution._dist for m in mc.models]
    assert all(d._resident is not None for d in dists)
    w = es.LogLikelihoodWeight()(mc, obs, standardisation_constant=1e-3)
    bary = es.Barycentre()(mc, w)
    wt = es.ModelSimilarityWeight()(mc, mode="temporal")
    ws = es.ModelSimilarityWeight()(mc)
    # nothing of O(N^2) has been downloaded so far
    assert all(d._cov_h is None and d._chol_h is None for d in dists)
    mus, covs, w_ref, bm, bv = _oracle_path(mc, obs, 2, 1e-3)
    assert relerr(w.values, w_ref) < 1e-8 and relerr(bary.mu, bm) < 1e-8 and relerr(bary.covariance, bv) < 1e-8
    var = np.stack([np.diag(c) for c in covs])
    assert relerr(wt.values, go.w2_weights_temporal(np.stack(mus), var)) < 1e-8
    assert relerr(ws.values[:, 0], go.w2_weights_single(mus, covs)[0]) < 1e-7
    # host access on demand: the reference's attributes
    for m, mo, co in zip(mc.models, mus, covs):
        d = m.distribution
        assert relerr(d.mu, mo) < 1e-9 and relerr(d.covariance, co) < 1e-9
        assert relerr(d.mean.values, mo) < 1e-9 and relerr(d.variance.values, np.diag(co)) < 1e-9
        assert relerr(d._dist._scale_tri, np.linalg.cholesky(co)) < 1e-9


def test_fp32_fit_resident_tables_are_float64(ctx):
    mc, obs = _collection((4, 4, 4))
    mc.fit(model=es.GPDTW1D(dtype="float32"), n_optim_nits=2, progress_bar=False)
    w = es.LogLikelihoodWeight()(mc, obs, standardisation_constant=1e-3)
    bary = es.Barycentre()(mc, w)
    _, _, w_ref, bm, bv = _oracle_path(mc, obs, 2, 1e-3)
    assert relerr(w.values, w_ref) < 1e-3 and relerr(bary.mu, bm) < 1e-3 and relerr(bary.covariance, bv) < 1e-3
    assert mc[0].distribution.covariance.dtype == np.float64


def test_pickle_takes_the_posterior_to_the_host(ctx, tmp_path):
    mc, obs = _collection((3, 3))
    mc.fit(model=es.GPDTW1D(), n_optim_nits=3)
    w0 = es.LogLikelihoodWeight()(mc, obs, standardisation_constant=1e-3)
    mc.save(str(tmp_path / "mc.pkl"))                                   # data.py:397-399
    back = pickle.load(open(tmp_path / "mc.pkl", "rb"))
    for a, b in zip(mc.models, back.models):
        assert b.distribution._dist._resident is None
        assert np.array_equal(a.distribution.mu, b.distribution.mu)
        assert np.array_equal(a.distribution.covariance, b.distribution.covariance)
        assert np.array_equal(a.distribution.losses, b.distribution.losses)
    w1 = es.LogLikelihoodWeight()(back, obs, standardisation_constant=1e-3)
    assert np.array_equal(w0.values, w1.values)


@pytest.mark.parametrize("N", [40, 130])
def test_general_log_prob(ctx, N):
    """distrax log_prob for ordinary arguments: (N,) -> scalar, (R, N) -> (R,), resident and host-constructed alike."""
    mc, _ = _collection((5,), N=N)
    mc.fit(model=es.GPDTW1D(), n_optim_nits=2, progress_bar=False)
    d = mc[0].distribution._dist
    X, y, nu = go.prologue(mc[0].model_data.values)
    mo, co, _ = go.gp_fit(X, y, nu, 2)
    Lc, ld = go.mvn_factor(co)
    rng = np.random.default_rng(1)
    x = mo + 0.2 * rng.standard_normal((70, N))                          # 70 arguments: two passes of 64 rows
    ref = go.mvn_logprob(mo, Lc, ld, x)
    got = d.log_prob(x)
    assert got.shape == (70,) and np.allclose(got, ref, rtol=1e-9, atol=1e-7)
    assert d._chol_h is None                                            # the factor stayed in HBM
    assert abs(d.log_prob(x[3]) - ref[3]) < 1e-7 * max(1.0, abs(ref[3]))
    host = dx.MultivariateNormalFullCovariance(mo, co)
    assert np.allclose(host.log_prob(x[:5]), ref[:5], rtol=1e-9, atol=1e-7)
    # the (T, 1) broadcast form of the reference's weight is unchanged
    t = np.linspace(-1, 2, 7)[:, None]
    assert np.allclose(d.log_prob(t), go.mvn_constvec_loglik(mo, Lc, ld, t[:, 0][None])[0], rtol=1e-9, atol=1e-7)
