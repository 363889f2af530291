"""The fused small-member fit kernel (csrc/small.hip; BASELINE.json config 5: per-gridpoint fits at the reference's
annual horizon N = 86, experiments/full_experiment_script.py:57,70) against the oracle.  Every fit with N <= 94 that goes
through bayens_gp_fit_batch takes this kernel, so tests/test_gpu_fit_parity.py covers it at the small shapes too; here:
the batch scale of config 5, the reduced export, the boundaries of the tile grid and the general path as cross-check."""
import numpy as np
import pytest
from threadpoolctl import threadpool_limits

from bayesian_ensembling_amd import _lib as L
from conftest import member_inputs, relerr
from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu


def _batch(B, N, R, seed):
    """B independent members (one per grid point in config 5): quadratic trend + integrated noise, seeded."""
    rng = np.random.default_rng(seed)
    tau = np.linspace(0, 1, N)
    amp = 0.8 + 0.4 * rng.random((B, 1, 1))
    data = 1.5 * amp * tau ** 2 + 0.036 * rng.standard_normal((B, R, N)).cumsum(axis=2)
    return member_inputs(data)


def test_config5_batch_scale(ctx):
    """B = 8192 members of N = 86 (1/54 of config 5's 442 368 point-fits): every member factorises (info == 0), a
    sample of members matches the oracle to 1e-10, the outputs are exactly symmetric / lower triangular."""
    B, N, R, T = 8192, 86, 8, 2
    X, y, nu = _batch(B, N, R, seed=5)
    dX, dy, dnu = ctx.to_device(X), ctx.to_device(y), ctx.to_device(nu)
    dmu, dcov, dchol = ctx.empty((B, N)), ctx.empty((B, N, N)), ctx.empty((B, N, N))
    dld, dhyp, dbasis = ctx.empty((B,)), ctx.empty((B, 2)), ctx.empty((B, 2, N))
    bufs = [dX, dy, dnu, dmu, dcov, dchol, dld, dhyp, dbasis]
    try:
        info = ctx.gp_fit_batch_dev(B, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis)
        assert not info.any()
        mu, cov, chol = dmu.download(), dcov.download(), dchol.download()
        ld, hyp, basis = dld.download(), dhyp.download(), dbasis.download()
    finally:
        for b in bufs:
            b.free()
    assert np.isfinite(cov).all() and np.isfinite(mu).all()
    assert np.array_equal(cov, np.swapaxes(cov, 1, 2))
    assert not np.triu(chol, 1).any()
    assert relerr(np.einsum("bij,bkj->bik", chol[::512], chol[::512]), cov[::512]) < 1e-12      # chol chol^T == cov
    with threadpool_limits(1):
        for m in list(range(0, B, 683)) + [B - 1]:        # 13 members spread over the grid's workgroups
            mo, co, ho = go.gp_fit(X[m], y[m], nu[m], T)
            Lc, ldo = go.mvn_factor(co)
            assert relerr(mu[m], mo) < 1e-10 and relerr(cov[m], co) < 1e-10 and relerr(chol[m], Lc) < 1e-10, m
            assert relerr(hyp[m], ho) < 1e-10 and abs(ld[m] - ldo) < 1e-9 * abs(ldo), m
    # reduced export (mu, diag, logdet, hyp, basis) of the same members: bit-identical to the full export's
    red = ctx.gp_fit_batch_reduced(X[:300], y[:300], nu[:300], T)
    assert np.array_equal(red["mu"], mu[:300]) and np.array_equal(red["var"], np.einsum("bii->bi", cov[:300]))
    assert np.array_equal(red["logdet"], ld[:300]) and np.array_equal(red["hyp"], hyp[:300])
    assert np.array_equal(red["basis"], basis[:300])


@pytest.mark.parametrize("N,R,T,kernel", [(94, 8, 2, 0), (93, 3, 1, 0), (62, 4, 2, 0), (63, 4, 2, 0), (30, 5, 3, 0), (31, 5, 1, 1),
                                          (16, 2, 2, 0), (14, 9, 2, 2), (86, 25, 2, 0), (86, 3, 40, 0)])
def test_tile_grid_boundaries_against_the_oracle(ctx, N, R, T, kernel):
    """N at the edges of the 2 / 4 / 6-tile grids (30, 62 and 94 leave exactly the two padded rows the bordered Cholesky
    needs), R not a multiple of 4 and above 8, 40 iterations - each against the oracle."""
    data, _ = go.synthetic_ensemble(3, N, R=R, seed=100 + N + R)
    X, y, nu = member_inputs(data)
    out = ctx.gp_fit_batch(X, y, nu, T, kernel=kernel)
    assert not out["info"].any()
    tight = 1e-10 if kernel == 0 else 1e-6
    with threadpool_limits(1):
        for m in range(3):
            mo, co, ho = go.gp_fit(X[m], y[m], nu[m], T, kernel=kernel)
            Lc, ld = go.mvn_factor(co)
            assert relerr(out["mu"][m], mo) < tight and relerr(out["cov"][m], co) < tight and relerr(out["chol"][m], Lc) < tight
            assert relerr(out["hyp"][m], ho) < tight and abs(out["logdet"][m] - ld) < 1e-9 * max(1.0, abs(ld))


def test_small_and_general_paths_agree(ctx):
    """The same member through the fused kernel (N = 94) and, with one more time point appended, through the general
    path (N = 95): two different programs for the same arithmetic - both must sit on the oracle."""
    data, _ = go.synthetic_ensemble(2, 95, R=6, seed=77)
    X, y, nu = member_inputs(data)
    big = ctx.gp_fit_batch(X, y, nu, 2)
    small = ctx.gp_fit_batch(X[:, :94], y[:, :94], nu[:, :94], 2)
    for m in range(2):
        mo, co, _ = go.gp_fit(X[m], y[m], nu[m], 2)
        ms, cs, _ = go.gp_fit(X[m, :94], y[m, :94], nu[m, :94], 2)
        assert relerr(big["cov"][m], co) < 1e-10 and relerr(small["cov"][m], cs) < 1e-10
        assert relerr(big["mu"][m], mo) < 1e-10 and relerr(small["mu"][m], ms) < 1e-10


def test_failed_pivot_is_reported_per_member(ctx):
    data, _ = go.synthetic_ensemble(3, 40, R=3, seed=1)
    X, y, nu = member_inputs(data)
    nu[1, :] = 0.0                    # zero variance across realisations: 1 / nu is infinite (the reference fails too)
    with pytest.raises(L.BayensError) as e:
        ctx.gp_fit_batch(X, y, nu, 1)
    assert e.value.code == -2 and e.value.info[1] > 0 and e.value.info[0] == 0 and e.value.info[2] == 0


def test_reduced_export_of_large_members_through_the_general_path(ctx):
    """N > 254 (round 5; rounds 1-4 refused it): the reduced export comes from the general path - the same mu / diag(cov) /
    logdet / hypers / basis as the full export of the same fit, bit for bit, with no N x N table leaving the device."""
    data, _ = go.synthetic_ensemble(3, 255, R=4, seed=255)
    X, y, nu = member_inputs(data)
    red = ctx.gp_fit_batch_reduced(X, y, nu, 2)
    full = ctx.gp_fit_batch(X, y, nu, 2)
    assert not red["info"].any()
    assert np.array_equal(red["mu"], full["mu"]) and np.array_equal(red["var"], np.diagonal(full["cov"], axis1=1, axis2=2))
    for key in ("logdet", "hyp", "basis"):
        assert np.array_equal(red[key], full[key]), key


def test_reduced_export_of_mid_size_members_through_the_fused_mid_size_kernel(ctx):
    """94 < N <= 254 (the reference's historical annual horizon N = 165, experiments/full_experiment_script.py:57): the
    reduced export comes from csrc/medium.hip - bit-identical to the full export's mu / diag(cov) / logdet / hypers / basis
    of the same fit, and on the oracle."""
    import scipy.linalg as sla
    for (M, N, R, T) in [(3, 165, 9, 2), (2, 95, 4, 1), (2, 254, 5, 2)]:
        data, _ = go.synthetic_ensemble(M, N, R=R, seed=N)
        X, y, nu = member_inputs(data)
        red = ctx.gp_fit_batch_reduced(X, y, nu, T)
        ctx.set_medium_min_batch(1)
        try:
            full = ctx.gp_fit_batch(X, y, nu, T)
        finally:
            ctx.set_medium_min_batch(0)
        assert not red["info"].any()
        assert np.array_equal(red["mu"], full["mu"]) and np.array_equal(red["var"], np.einsum("bii->bi", full["cov"]))
        assert np.array_equal(red["logdet"], full["logdet"]) and np.array_equal(red["hyp"], full["hyp"])
        assert np.array_equal(red["basis"], full["basis"])
        with threadpool_limits(2):
            mo, co, ho = go.gp_fit(X[0], y[0], nu[0], T)
        Lc, ld = go.mvn_factor(co)
        assert relerr(red["mu"][0], mo) < 1e-10 and relerr(red["var"][0], np.diag(co)) < 1e-10 and relerr(red["hyp"][0], ho) < 1e-10
        assert abs(red["logdet"][0] - ld) < 1e-9 * abs(ld)
        assert relerr(red["basis"][0, 1], sla.solve_triangular(Lc, mo, lower=True)) < 1e-9


def test_fp32_instance_at_batch_scale(ctx):
    """The fp32 build of the fused kernel (fp32 tiles and MFMA products; diagonal factorisations, squared distances,
    vectors and reductions in fp64): B = 2048 members of N = 86 against the fp64 oracle at BASELINE's 1e-3."""
    B, N, R, T = 2048, 86, 8, 2
    X, y, nu = _batch(B, N, R, seed=9)
    out = ctx.gp_fit_batch(X, y, nu, T, dtype=np.float32, want_loss=True)
    assert not out["info"].any() and out["cov"].dtype == np.float32
    assert np.isfinite(out["cov"]).all() and np.array_equal(out["cov"], np.swapaxes(out["cov"], 1, 2))
    with threadpool_limits(1):
        for m in list(range(0, B, 256)) + [B - 1]:
            mo, co, ho, lo = go.gp_fit(X[m], y[m], nu[m], T, loss_every=25)
            Lc, ld = go.mvn_factor(co)
            assert relerr(out["mu"][m].astype(np.float64), mo) < 1e-3 and relerr(out["cov"][m].astype(np.float64), co) < 1e-3, m
            assert relerr(out["chol"][m].astype(np.float64), Lc) < 1e-3 and relerr(out["hyp"][m], ho) < 1e-3, m
            assert abs(out["logdet"][m] - ld) < 1e-3 * abs(ld) and np.allclose(out["loss"][m], lo, rtol=1e-3), m


def test_config5_weighting_step_over_the_chunk_boundary(ctx):
    """BASELINE config 5's weighting half, which bench.py times: a field of posteriors [M][G] (b = m * G + g) ->
    bayens_loglik_rows_basis_grouped (one observation set per grid point: `b % n_groups`) -> normalisation over the
    members -> barycentre, at B = 32 * 1100 = 35 200 > 32 768, so that the second chunk of the row kernel's grid runs
    (kernels.hip: `b0` loop) - against the oracle PER GRID POINT on a sample of points that includes the last posterior
    of chunk 1 (b = 32 767: member 29, point 867) and the first of chunk 2 (point 868).
    Reference: ensembles/weights.py:88-123 (rows, exp, normalisation over 'model'), ensemble_scheme.py:52-78 (space
    flattened into the point axis), wasserstein.py:61-100."""
    M, G, N, R, R_obs, T = 32, 1100, 30, 4, 3, 2
    B, P = M * G, G * N
    rng = np.random.default_rng(0xC5)
    tau = np.linspace(0.0, 1.0, N)
    amp = 0.8 + 0.4 * rng.random((M, 1, 1, 1))
    loc = 1.0 + 0.2 * rng.standard_normal((1, G, 1, 1))
    data = (1.5 * amp * loc * tau ** 2 + 0.036 * rng.standard_normal((M, G, R, N)).cumsum(axis=3)).reshape(B, R, N)
    obs = 1.5 * loc[0] * tau ** 2 + 0.036 * rng.standard_normal((G, R_obs, N)).cumsum(axis=2)      # [G][R_obs][N]
    X, y, nu = member_inputs(data)
    red = ctx.gp_fit_batch_reduced(X, y, nu, T)
    assert not red["info"].any()
    ll = ctx.loglik_rows_basis_grouped(red["basis"], red["logdet"], obs)
    assert ll.shape == (B, N) and np.isfinite(ll).all()
    c_std = 1e-2
    w = ctx.weights_normalise(ll.reshape(M, P), c_std)
    bmu, bvar, bad = ctx.barycentre(w, red["mu"].reshape(M, P), red["var"].reshape(M, P))
    assert bad == 0 and np.allclose(w.sum(axis=0), 1.0, rtol=1e-12)
    assert (32767 // G, 32767 % G) == (29, 867)
    with threadpool_limits(1):
        for g in (0, 1, 500, 867, 868, 869, G - 1):
            mus, vars_, lls = [], [], []
            for m in range(M):
                b = m * G + g
                mo, co, _ = go.gp_fit(X[b], y[b], nu[b], T)
                Lc, ld = go.mvn_factor(co)
                lls.append(go.loglik_mean_rows(go.mvn_constvec_loglik(mo, Lc, ld, obs[g])))
                mus.append(mo); vars_.append(np.diag(co))
                assert relerr(red["mu"][b], mo) < 1e-10 and relerr(red["var"][b], vars_[-1]) < 1e-10, (g, m)
            lls, mus, vars_ = np.stack(lls), np.stack(mus), np.stack(vars_)
            assert relerr(ll.reshape(M, G, N)[:, g], lls) < 1e-10, g
            w_ref, _ = go.loglik_weights(lls, c_std)
            assert relerr(w.reshape(M, G, N)[:, g], w_ref) < 1e-9, g
            rm, rv, _ = go.barycentre(w_ref, mus, vars_)
            assert relerr(bmu.reshape(G, N)[g], rm) < 1e-9 and relerr(bvar.reshape(G, N)[g], rv) < 1e-9, g


def test_config5_at_the_benchmarked_launch_shape(ctx):
    """bench.py's own config-5 launch (FieldWorkload(ctx, 32, 1728, 86, 8, T = 2, "reduced"), built by bench.py's class on
    bench.py's seed): B = 55 296 fits of N = 86 in ONE launch of the fused kernel with the reduced export on resident buffers.
    info == 0 for every fit; 13 members spread over the launch (first / last workgroup round included) against the oracle; the
    per-gridpoint weights and barycentre of the step at three grid points against weights.py:88-123 / ensemble_scheme.py:52-78."""
    import bench
    M, G, N, R, T = 32, 1728, 86, 8, 2
    fw = bench.FieldWorkload(ctx, M, G, N, R, T, "reduced", keep_host=True)
    try:
        X, y, nu, obs = fw.host
        B = fw.B
        assert B == 55296
        info = fw.fit()
        assert info.shape == (B,) and not info.any()
        bad = fw.weighted_posterior()
        mu, var, ld, hyp = fw.dmu.download(), fw.dvar.download(), fw.dld.download(), fw.dhyp.download()
        basis, w = fw.dbasis.download(), fw.dw.download().reshape(M, G, N)
        bmu, bvar = fw.dbmu.download().reshape(G, N), fw.dbvar.download().reshape(G, N)
    finally:
        fw.free()
    assert bad == 0 and np.isfinite(mu).all() and np.isfinite(var).all() and np.isfinite(ld).all() and np.isfinite(basis).all()
    import scipy.linalg as sla
    with threadpool_limits(1):
        for b in list(range(0, B, 4608)) + [B - 1]:                 # 13 members
            mo, co, ho = go.gp_fit(X[b], y[b], nu[b], T)
            Lc, ldo = go.mvn_factor(co)
            assert relerr(mu[b], mo) < 1e-10 and relerr(var[b], np.diag(co)) < 1e-10 and relerr(hyp[b], ho) < 1e-10, b
            assert abs(ld[b] - ldo) < 1e-9 * abs(ldo), b
            assert relerr(basis[b, 0], sla.solve_triangular(Lc, np.ones(N), lower=True)) < 1e-9, b
            assert relerr(basis[b, 1], sla.solve_triangular(Lc, mo, lower=True)) < 1e-9, b
        for g in (0, 1000, G - 1):
            lls, mus, vars_ = [], [], []
            for m in range(M):
                b = m * G + g
                mo, co, _ = go.gp_fit(X[b], y[b], nu[b], T)
                Lc, ldo = go.mvn_factor(co)
                lls.append(go.loglik_mean_rows(go.mvn_constvec_loglik(mo, Lc, ldo, obs[g])))
                mus.append(mo); vars_.append(np.diag(co))
            w_ref, _ = go.loglik_weights(np.stack(lls), 1e-3)
            assert np.isfinite(w_ref).all()
            big = w_ref > 1e-300
            assert np.max(np.abs(w[:, g][big] / w_ref[big] - 1.0)) < 1e-5, g          # element-wise (north_star: weights to 1e-5)
            rm, rv, _ = go.barycentre(w_ref, np.stack(mus), np.stack(vars_))
            assert relerr(bmu[g], rm) < 1e-9 and relerr(bvar[g], rv) < 1e-9, g
