"""The fused mid-size fit kernel (csrc/medium.hip): members with 94 < N <= 254 time points - the reference's real workload,
historical annual series N = 165 fitted with n_optim_nits = 2000-2500 (experiments/full_experiment_script.py:57,87-113) -
one workgroup per member, matrices in an L2-resident scratch.  bayens_gp_fit_batch takes it by size (N <= 190 always, above
that for large batches); here it is forced for every size and held to the oracle, to the reference's own pickled
N = 165 posteriors, and to the general path."""
import os

import numpy as np
import pytest
from threadpoolctl import threadpool_limits

from conftest import GOLDEN, golden_files, member_inputs, relerr
from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu


@pytest.fixture()
def mctx(ctx):
    ctx.set_medium_min_batch(1)
    yield ctx
    ctx.set_medium_min_batch(0)


@pytest.mark.parametrize("M,N,R,T,kernel", [
    (2, 95, 4, 2, 0), (3, 130, 4, 2, 0), (2, 165, 9, 3, 0), (1, 174, 5, 2, 0), (1, 175, 5, 1, 0), (2, 200, 8, 2, 0),
    (1, 254, 6, 2, 0), (2, 130, 4, 2, 1), (1, 100, 8, 2, 2), (1, 128, 4, 51, 0), (1, 200, 8, 0, 0), (1, 158, 50, 1, 0)])
def test_medium_kernel_matches_the_oracle(mctx, M, N, R, T, kernel):
    """Both ends of the range, the edges of the tile grid (174 | 175: 11 | 12 tiles of 16 with the two bordering rows), the
    extension kernels, loss samples over more than 25 iterations, n_iters = 0, many realisations."""
    data, _ = go.synthetic_ensemble(M, N, R=R, seed=N + R)
    X, y, nu = member_inputs(data)
    out = mctx.gp_fit_batch(X, y, nu, T, kernel=kernel, want_loss=True)
    assert not out["info"].any()
    tight = 1e-10 if kernel == 0 else 1e-6
    with threadpool_limits(2):
        for m in range(M):
            res = go.gp_fit(X[m], y[m], nu[m], T, kernel=kernel, loss_every=25) if T > 0 else go.gp_fit(X[m], y[m], nu[m], T, kernel=kernel)
            mo, co, ho = res[:3]
            Lc, ld = go.mvn_factor(co)
            assert relerr(out["mu"][m], mo) < tight and relerr(out["cov"][m], co) < tight
            assert relerr(out["chol"][m], Lc) < tight and relerr(out["hyp"][m], ho) < tight
            assert abs(out["logdet"][m] - ld) < 1e-9 * max(1.0, abs(ld))
            assert np.array_equal(out["cov"][m], out["cov"][m].T) and not np.triu(out["chol"][m], 1).any()
            if T > 0:
                assert np.allclose(out["loss"][m], res[3], rtol=1e-9)
            import scipy.linalg as sla
            assert relerr(out["basis"][m, 0], sla.solve_triangular(Lc, np.ones(N), lower=True)) < 10 * tight
            assert relerr(out["basis"][m, 1], sla.solve_triangular(Lc, mo, lower=True)) < 10 * tight
    again = mctx.gp_fit_batch(X, y, nu, T, kernel=kernel, want_loss=True)        # reproducible bit for bit, loss samples included
    for key in ("mu", "cov", "chol", "logdet", "hyp", "basis") + (("loss",) if T > 0 else ()):
        assert np.array_equal(out[key], again[key]), key


@pytest.mark.parametrize("fname", [f for f in golden_files() if "hist" in f])
def test_medium_kernel_reproduces_the_references_pickled_n165_posterior(mctx, fname):
    """experiments/pre_fit_models/histssp*.pkl (N = 165, GPDTW1D with n_optim_nits = 2500) through the fused mid-size kernel."""
    g = np.load(os.path.join(GOLDEN, fname))
    X, _, nu = go.prologue(g["data"])
    assert 94 < X.shape[0] <= 254
    out = mctx.gp_fit_batch(X[None], g["y_mean"][None], nu[None], int(g["n_iters"]))
    assert relerr(out["cov"][0], g["cov_gold"]) < 1e-6
    assert np.max(np.abs(out["mu"][0] - g["mu_gold"])) < 1e-5
    assert relerr(out["hyp"][0], g["hyp"]) < 1e-5


def test_the_papers_collection_shape_every_member_against_the_oracle(mctx):
    """20 members x N = 165 x R = 9 (pre_fitting_cmip6models.py:76-77: one collection of the paper) in ONE batch, every member and
    every output - the log-likelihood basis included - against the oracle.  This is the shape on which a build of round 4 faulted:
    the compiler had parked a register across the call of the factorisation with EXEC = 0 (DESIGN section 5c;
    tools/check_exec_prologue.py keeps that out of the build), `e / N` of the basis export came out as garbage and the kernel read
    below its scratch - with 20 members the scratch is a mapping of its own and the read faults, with 1-2 members it only
    returned a wrong basis, which the diagnostic of that round did not look at."""
    import scipy.linalg as sla
    M, N, R, T = 20, 165, 9, 2
    data, _ = go.synthetic_ensemble(M, N, R=R, seed=N + R)
    X, y, nu = member_inputs(data)
    out = mctx.gp_fit_batch(X, y, nu, T)
    assert not out["info"].any()
    with threadpool_limits(4):
        for m in range(M):
            mo, co, ho = go.gp_fit(X[m], y[m], nu[m], T)[:3]
            Lc, ld = go.mvn_factor(co)
            assert relerr(out["mu"][m], mo) < 1e-10 and relerr(out["cov"][m], co) < 1e-10, m
            assert relerr(out["chol"][m], Lc) < 1e-10 and relerr(out["hyp"][m], ho) < 1e-10, m
            assert abs(out["logdet"][m] - ld) < 1e-9 * max(1.0, abs(ld)), m
            assert relerr(out["basis"][m, 0], sla.solve_triangular(Lc, np.ones(N), lower=True)) < 1e-9, m
            assert relerr(out["basis"][m, 1], sla.solve_triangular(Lc, mo, lower=True)) < 1e-9, m
    red = mctx.gp_fit_batch_reduced(X, y, nu, T)                       # the reduced export of the same batch (per-gridpoint fields)
    assert np.array_equal(red["mu"], out["mu"]) and np.array_equal(red["basis"], out["basis"]) and np.array_equal(red["hyp"], out["hyp"])


def test_medium_and_general_paths_agree_and_large_batches_take_the_fused_kernel(ctx):
    """A batch of 130 members of N = 120: the default dispatch (fused kernel) against the general path (forced), member by
    member; members at the edges of the grid of workgroups against the oracle.  And N = 200 (13 tiles per side): general
    path for a small batch by default, fused kernel when forced - both on the oracle."""
    M, N, R, T = 130, 120, 4, 2
    data, _ = go.synthetic_ensemble(M, N, R=R, seed=7)
    X, y, nu = member_inputs(data)
    fused = ctx.gp_fit_batch(X, y, nu, T)
    ctx.set_medium_min_batch(10 ** 6)
    try:
        general = ctx.gp_fit_batch(X, y, nu, T)
    finally:
        ctx.set_medium_min_batch(0)
    for k in ("mu", "cov", "chol", "hyp", "basis"):
        assert relerr(fused[k], general[k]) < 1e-10, k
    assert not np.array_equal(fused["cov"], general["cov"])          # two different programs (else the dispatch did not switch)
    with threadpool_limits(2):
        for m in (0, 1, 64, 128, 129):
            mo, co, ho = go.gp_fit(X[m], y[m], nu[m], T)
            assert relerr(fused["mu"][m], mo) < 1e-10 and relerr(fused["cov"][m], co) < 1e-10 and relerr(fused["hyp"][m], ho) < 1e-10
    data, _ = go.synthetic_ensemble(2, 200, R=4, seed=8)
    X, y, nu = member_inputs(data)
    by_default = ctx.gp_fit_batch(X, y, nu, T)
    ctx.set_medium_min_batch(1)
    try:
        forced = ctx.gp_fit_batch(X, y, nu, T)
    finally:
        ctx.set_medium_min_batch(0)
    assert relerr(forced["cov"], by_default["cov"]) < 1e-10 and not np.array_equal(forced["cov"], by_default["cov"])


def test_more_members_than_workgroups(mctx):
    """300 members of N = 96 on 256 CUs: workgroups take a second member and re-initialise their scratch."""
    M, N, R, T = 300, 96, 3, 1
    data, _ = go.synthetic_ensemble(M, N, R=R, seed=11)
    X, y, nu = member_inputs(data)
    out = mctx.gp_fit_batch(X, y, nu, T, want_chol=False)
    assert not out["info"].any()
    with threadpool_limits(2):
        for m in (0, 255, 256, 299):
            mo, co, _ = go.gp_fit(X[m], y[m], nu[m], T)
            assert relerr(out["mu"][m], mo) < 1e-10 and relerr(out["cov"][m], co) < 1e-10


def test_plugin_api_takes_the_fused_kernel_for_the_papers_series_length(ctx):
    """ModelCollection.fit(GPDTW1D) on members of the paper's length (N = 165 is what experiments/full_experiment_script.py
    fits; 3 members x N = 130 here to keep the oracle cheap): resident posteriors produced by the fused mid-size kernel,
    LogLikelihoodWeight and Barycentre on them, against the oracle; loss samples kept with progress_bar=True."""
    import bayesian_ensembling_amd as es
    from bayesian_ensembling_amd.labelled import DataArray
    M, N, R, T = 3, 130, 5, 27
    data, obs = go.synthetic_ensemble(M, N, R=R, R_obs=2, seed=165)
    time = np.arange(N)
    pm = lambda a, name: es.ProcessModel(DataArray(a, ("realisation", "time"), {"realisation": np.arange(a.shape[0]), "time": time}), name)
    mc = es.ModelCollection([pm(data[m], f"m{m}") for m in range(M)])
    mc.fit(model=es.GPDTW1D(), n_optim_nits=T)                        # progress_bar=True: two loss samples
    w = es.LogLikelihoodWeight()(mc, pm(obs, "obs"), standardisation_constant=1e-3)
    post = es.Barycentre()(mc, w)
    lls, mus, vs = [], [], []
    with threadpool_limits(2):
        for m in range(M):
            X, y, nu = go.prologue(data[m])
            mo, co, _, lo = go.gp_fit(X, y, nu, T, loss_every=25)
            assert relerr(mc[m].distribution.mean.values, mo) < 1e-9 and relerr(np.asarray(mc[m].distribution.covariance), co) < 1e-9
            assert np.allclose(mc[m].distribution.losses, lo, rtol=1e-9)
            lls.append(go.loglik_mean_rows(go.mvn_constvec_loglik(mo, *go.mvn_factor(co), obs)))
            mus.append(mo); vs.append(np.diag(co))
    w_ref, _ = go.loglik_weights(np.stack(lls), 1e-3)
    assert relerr(w.values, w_ref) < 1e-8
    rm, rv, _ = go.barycentre(w_ref, np.stack(mus), np.stack(vs))
    assert relerr(post.mu, rm) < 1e-8 and relerr(post.covariance, rv) < 1e-8
