"""Members whose realisation counts differ, and several collections, in ONE launch (bayens_gp_fit_batch_ragged).
The reference fits the members of a collection one after the other (ensembles/data.py:385-395) and a script's collections one
after the other (experiments/full_experiment_script.py:87-113); its own pickled collections hold R in {25, 9, 16}, {25, 9, 10},
{25, 25, 5, 6, 3, 5}.  Every member of equal length goes into one batch here, and each member's result must be bit for bit what a
uniform batch of its own R gives - on all three routes (fused small, fused mid-size, general)."""
import os

import numpy as np
import pytest
from threadpoolctl import threadpool_limits

from conftest import GOLDEN, golden_files, golden_tolerances, relerr
from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu
KEYS = ("mu", "cov", "chol", "logdet", "hyp", "basis")


def ragged_members(N, Rs, seed):
    datas = [go.synthetic_ensemble(1, N, R=r, seed=seed + 7 * k + r)[0][0] for k, r in enumerate(Rs)]       # (R, N) each
    return [np.ascontiguousarray(d.T) for d in datas], np.stack([d.mean(axis=0) for d in datas]), np.stack([d.var(axis=0) for d in datas])


@pytest.mark.parametrize("N,Rs,T,route", [
    (86, [25, 25, 5, 6, 3, 5, 2, 9], 3, "small"), (40, [2, 50, 8], 2, "small"),
    (165, [25, 9, 16, 10, 3, 50], 3, "medium"), (120, [4, 9, 4], 28, "medium"),
    (165, [25, 9, 16, 3], 3, "general"), (300, [8, 3, 13, 5, 8], 2, "general")])
def test_ragged_batch_is_bit_for_bit_the_members_own_uniform_batches(ctx, N, Rs, T, route):
    Xs, y, nu = ragged_members(N, Rs, seed=N)
    ctx.set_fit_route(1 if route == "general" else 0)
    try:
        out = ctx.gp_fit_batch_ragged(Xs, y, nu, T, want_loss=True)
        assert not out["info"].any()
        for r in sorted(set(Rs)):
            idx = [k for k, rr in enumerate(Rs) if rr == r]
            ref = ctx.gp_fit_batch(np.stack([Xs[k] for k in idx]), y[idx], nu[idx], T, want_loss=True)
            for key in KEYS + ("loss",):
                assert np.array_equal(out[key][idx], ref[key]), (key, r)
    finally:
        ctx.set_fit_route(0)
    with threadpool_limits(2):                       # and two members against the oracle
        for m in (0, len(Rs) - 1):
            mo, co, ho = go.gp_fit(Xs[m], y[m], nu[m], T)[:3]
            assert relerr(out["mu"][m], mo) < 1e-9 and relerr(out["cov"][m], co) < 1e-9 and relerr(out["hyp"][m], ho) < 1e-9


def test_explicit_offsets_blocks_out_of_order_with_gaps_and_two_host_groups(ctx):
    """The C entry with x_off given: blocks of X in another order than the members, NaN-filled gaps between them; the general
    route with its members split over two host groups (each group's view of the layout arrays is offset, X's base is not)."""
    N, T, Rs = 200, 2, [6, 3, 9, 4, 5]
    Xs, y, nu = ragged_members(N, Rs, seed=3)
    order, pos, chunks = [3, 0, 4, 2, 1], 0, []
    off = np.zeros(len(Rs), dtype=np.uint64)
    for k in order:
        chunks.append(np.full(11, np.nan)); pos += 11
        off[k] = pos
        chunks.append(Xs[k].ravel()); pos += Xs[k].size
    X = np.concatenate(chunks)
    B = len(Rs)
    ref = ctx.gp_fit_batch_ragged(Xs, y, nu, T)
    dX, dy, dnu = ctx.to_device(X), ctx.to_device(y), ctx.to_device(nu)
    dmu, dcov, dchol = ctx.empty((B, N)), ctx.empty((B, N, N)), ctx.empty((B, N, N))
    dld, dhyp, dbasis = ctx.empty((B,)), ctx.empty((B, 2)), ctx.empty((B, 2, N))
    ctx.set_groups(2)
    try:
        info = ctx.gp_fit_batch_ragged_dev(B, N, np.array(Rs, dtype=np.int32), off, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis)
    finally:
        ctx.set_groups(0)
    assert not info.any()
    got = {"mu": dmu.download(), "cov": dcov.download(), "chol": dchol.download(), "logdet": dld.download(), "hyp": dhyp.download(),
           "basis": dbasis.download()}
    for key in KEYS:
        assert np.array_equal(got[key], ref[key]), key
    for d in (dX, dy, dnu, dmu, dcov, dchol, dld, dhyp, dbasis):
        d.free()


def test_bad_arguments_are_refused(ctx):
    from bayesian_ensembling_amd._lib import BayensError
    Xs, y, nu = ragged_members(30, [3, 4], seed=1)
    R = np.array([3, 0], dtype=np.int32)
    out = [np.empty((2, 30)), np.empty((2, 30, 30))]
    from bayesian_ensembling_amd import _lib as L
    rc = ctx.lib.bayens_gp_fit_batch_ragged(ctx.h, L.HOST, 2, 30, L._ptr(R), None, L._ptr(np.concatenate([x.ravel() for x in Xs])), L._ptr(y),
                                            L._ptr(nu), 0, 1, 0.5, 0.01, 1e-6, L._ptr(out[0]), L._ptr(out[1]), None, None, None, None, None, None)
    assert rc != 0 and b"at least one realisation" in ctx.lib.bayens_last_error(ctx.h)
    with pytest.raises(ValueError):
        ctx.ragged_layout([np.zeros((30, 3)), np.zeros((31, 3))])


def _fixture_batch(names, copies=1):
    Xs, y, nu, gold = [], [], [], []
    for _ in range(copies):
        for f in names:
            g = np.load(os.path.join(GOLDEN, f))
            X, _, v = go.prologue(g["data"])
            Xs.append(X); y.append(g["y_mean"]); nu.append(v); gold.append(g)
    return Xs, np.stack(y), np.stack(nu), gold


@pytest.mark.parametrize("which,copies", [("hist", 4), ("ssp", 1)])
def test_the_references_pickled_collections_as_one_launch_at_2500_iterations(ctx, which, copies):
    """experiments/pre_fit_models/*.pkl (GPDTW1D, n_optim_nits = 2500): the six N = 165 members (R = 25, 9, 16, 25, 9, 10) four times
    over - 24 workgroups of the fused mid-size kernel, the paper's collection size, pre_fitting_cmip6models.py:76-77 - and the twelve
    N = 86 members (R = 25 ... 3) as ONE ragged launch each, every member against the reference's own posterior (the bars of
    conftest.golden_tolerances), and the copies of a member bit-identical to each other."""
    names = [f for f in golden_files() if ("hist" in f) == (which == "hist")]
    Xs, y, nu, gold = _fixture_batch(names, copies)
    assert len({x.shape[1] for x in Xs}) > 2 and len(Xs) >= 12
    T = int(gold[0]["n_iters"])
    assert T == 2500
    out = ctx.gp_fit_batch_ragged(Xs, y, nu, T)
    assert not out["info"].any()
    for m, g in enumerate(gold):
        tc, tm, th = golden_tolerances(g)
        assert relerr(out["cov"][m], g["cov_gold"]) < tc, (m, names[m % len(names)])
        assert np.max(np.abs(out["mu"][m] - g["mu_gold"])) < tm, (m, names[m % len(names)])
        assert relerr(out["hyp"][m], g["hyp"]) < th, (m, names[m % len(names)])
        assert relerr(out["chol"][m], np.linalg.cholesky(out["cov"][m])) < 1e-9
        if m >= len(names):
            for key in KEYS:
                assert np.array_equal(out[key][m], out[key][m % len(names)]), (key, m)


def test_collections_with_ragged_members_through_the_plugin_api(ctx):
    """ModelCollection.fit on members of R = 4, 9, 4, 6 (one launch), and es.fit_collections on two collections at once: the same
    posteriors, bit for bit, as fitting collection by collection; weights and barycentre on the result against the oracle."""
    import bayesian_ensembling_amd as es
    from bayesian_ensembling_amd.labelled import DataArray
    N, T = 110, 4
    time = np.arange(N)
    pm = lambda a, name: es.ProcessModel(DataArray(a, ("realisation", "time"), {"realisation": np.arange(a.shape[0]), "time": time}), name)

    def collection(Rs, seed, n=N):
        t = np.arange(n)
        mk = lambda a, name: es.ProcessModel(DataArray(a, ("realisation", "time"), {"realisation": np.arange(a.shape[0]), "time": t}), name)
        return es.ModelCollection([mk(go.synthetic_ensemble(1, n, R=r, seed=seed + k)[0][0], f"m{seed}_{k}") for k, r in enumerate(Rs)])

    # (the second collection has ANOTHER length - a projection beside a historical period: its launch runs beside the first one's,
    #  through a side context)
    a1, b1 = collection([4, 9, 4, 6], 20), collection([3, 12, 5], 40, n=70)
    a2, b2 = collection([4, 9, 4, 6], 20), collection([3, 12, 5], 40, n=70)
    es.fit_collections([a1, b1], es.GPDTW1D(), n_optim_nits=T, progress_bar=False)
    a2.fit(model=es.GPDTW1D(), n_optim_nits=T, progress_bar=False)
    b2.fit(model=es.GPDTW1D(), n_optim_nits=T, progress_bar=False)
    for c1, c2 in ((a1, a2), (b1, b2)):
        for p1, p2 in zip(c1, c2):
            assert np.array_equal(p1.distribution.mean.values, p2.distribution.mean.values)
            assert np.array_equal(np.asarray(p1.distribution.covariance), np.asarray(p2.distribution.covariance))
    obs = go.synthetic_ensemble(1, N, R=4, R_obs=2, seed=99)[1]
    w = es.LogLikelihoodWeight()(a1, pm(obs, "obs"), standardisation_constant=1e-3)
    post = es.Barycentre()(a1, w)
    lls, mus, vs = [], [], []
    with threadpool_limits(2):
        for m in range(len(a1.models)):
            X, yv, nu = go.prologue(np.asarray(a1[m].model_data.values))
            mo, co, _ = go.gp_fit(X, yv, nu, T)[:3]
            assert relerr(a1[m].distribution.mean.values, mo) < 1e-9 and relerr(np.asarray(a1[m].distribution.covariance), co) < 1e-9
            lls.append(go.loglik_mean_rows(go.mvn_constvec_loglik(mo, *go.mvn_factor(co), obs)))
            mus.append(mo); vs.append(np.diag(co))
    w_ref, _ = go.loglik_weights(np.stack(lls), 1e-3)
    assert relerr(w.values, w_ref) < 1e-8
    rm, rv, _ = go.barycentre(w_ref, np.stack(mus), np.stack(vs))
    assert relerr(post.mu, rm) < 1e-8 and relerr(post.covariance, rv) < 1e-8


@pytest.mark.parametrize("N", [60, 300])
def test_ragged_collection_in_float32(ctx, N):
    """GPDTW1D(dtype="float32") on members of R = 4, 9, 4, 6: ONE launch of bayens_gp_fit_batch_ragged_f32 (fused small-member kernel at
    N = 60, the fp32 general path at N = 300) - bit for bit what uniform fp32 batches of each R give, and within BASELINE.json's fp32
    bar (1e-3 relative) of the float64 oracle."""
    import bayesian_ensembling_amd as es
    from bayesian_ensembling_amd.labelled import DataArray
    T, Rs = 3, [4, 9, 4, 6]
    t = np.arange(N)
    datas = [go.synthetic_ensemble(1, N, R=r, seed=300 + k)[0][0] for k, r in enumerate(Rs)]
    mk = lambda a, name: es.ProcessModel(DataArray(a, ("realisation", "time"), {"realisation": np.arange(a.shape[0]), "time": t}), name)
    mc = es.ModelCollection([mk(d, f"m{k}") for k, d in enumerate(datas)])
    mc.fit(model=es.GPDTW1D(dtype="float32"), n_optim_nits=T, progress_bar=False)
    for r in sorted(set(Rs)):
        idx = [k for k, rr in enumerate(Rs) if rr == r]
        X = np.stack([datas[k].T for k in idx])
        ref = ctx.gp_fit_batch(X, np.stack([datas[k].mean(axis=0) for k in idx]), np.stack([datas[k].var(axis=0) for k in idx]), T,
                               dtype=np.float32)
        for j, k in enumerate(idx):
            assert np.array_equal(np.asarray(mc[k].distribution.mean.values), ref["mu"][j].astype(np.float64)), (r, k)
            assert np.array_equal(np.asarray(mc[k].distribution.covariance), ref["cov"][j].astype(np.float64)), (r, k)
    with threadpool_limits(2):
        for k in (0, 1):
            X, yv, nu = go.prologue(datas[k])
            mo, co, _ = go.gp_fit(X, yv, nu, T)[:3]
            assert relerr(mc[k].distribution.mean.values, mo) < 1e-3 and relerr(np.asarray(mc[k].distribution.covariance), co) < 1e-3
