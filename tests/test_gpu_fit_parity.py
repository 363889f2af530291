"""The HIP fit vs the oracle on identical seeded inputs, the golden fixtures of the reference's own fits, and
size-independent properties at BASELINE.json's full size.  Tolerance: 1e-5 relative (fp64), from BASELINE.json;
the small cases are asserted much tighter."""
import os

import numpy as np
import pytest
from threadpoolctl import threadpool_limits

from bayesian_ensembling_amd import _lib as L
from conftest import GOLDEN, golden_files, golden_tolerances, member_inputs, relerr
from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu
REL_TOL = 1e-5


@pytest.mark.parametrize("M,N,R,T,kernel", [
    (8, 200, 8, 2, 0),      # BASELINE config 1: 8-member x 200-step, test-scale T (tests/test_weights.py:90)
    (2, 86, 5, 3, 0), (3, 24, 3, 2, 0),   # the reference test's own N=24, R=3
    (1, 65, 3, 1, 0), (1, 64, 2, 2, 0), (2, 130, 4, 2, 1), (1, 100, 8, 2, 2), (2, 7, 3, 2, 0), (1, 200, 8, 0, 0),
    # the extension kernels (Matern-5/2 = 1, RBF = 2; BASELINE configs 1-2 say "RBF GP") through the GENERAL path (N > 254):
    # one iteration (general form), three (symmetric first iteration with its gradient weights from P, then general ones)
    (2, 320, 6, 1, 1), (2, 320, 6, 3, 1), (2, 320, 6, 1, 2), (2, 320, 6, 3, 2), (1, 1000, 8, 2, 1), (1, 1000, 8, 2, 2)])
def test_fit_matches_oracle(ctx, M, N, R, T, kernel):
    data, _ = go.synthetic_ensemble(M, N, R=R)
    X, y, nu = member_inputs(data)
    out = ctx.gp_fit_batch(X, y, nu, T, kernel=kernel)
    assert not out["info"].any()
    # smoother kernels: K + 1e-6 I is far worse conditioned (1e-6 asserted up to N = 320; BASELINE's 1e-5 at N = 1000)
    tight = 1e-10 if kernel == 0 else (1e-6 if N <= 320 else REL_TOL)
    for m in range(M):
        mo, co, ho = go.gp_fit(X[m], y[m], nu[m], T, kernel=kernel)
        Lc, ld = go.mvn_factor(co)
        assert relerr(out["mu"][m], mo) < tight and relerr(out["cov"][m], co) < tight
        assert relerr(out["chol"][m], Lc) < tight and abs(out["logdet"][m] - ld) < 1e-6
        assert relerr(out["hyp"][m], ho) < tight
        assert np.array_equal(out["cov"][m], out["cov"][m].T)
        import scipy.linalg as sla
        assert relerr(out["basis"][m, 0], sla.solve_triangular(Lc, np.ones(N), lower=True)) < 10 * tight
        assert relerr(out["basis"][m, 1], sla.solve_triangular(Lc, mo, lower=True)) < 10 * tight


def test_fit_with_api_default_iterations(ctx):
    # n_optim_nits = 500 is the API default (models.py:167): the optimiser trajectory must stay on the oracle's
    data, _ = go.synthetic_ensemble(2, 86, R=5, seed=3)
    X, y, nu = member_inputs(data)
    out = ctx.gp_fit_batch(X, y, nu, 500)
    with threadpool_limits(1):
        for m in range(2):
            mo, co, ho = go.gp_fit(X[m], y[m], nu[m], 500)
            assert relerr(out["mu"][m], mo) < 1e-7 and relerr(out["cov"][m], co) < 1e-7 and relerr(out["hyp"][m], ho) < 1e-8


@pytest.mark.parametrize("fname", golden_files())
def test_fit_reproduces_the_references_pickled_posteriors(ctx, fname):
    """experiments/pre_fit_models/*.pkl -> tests/golden: GPDTW1D with n_optim_nits=2500 as the reference ran it."""
    g = np.load(os.path.join(GOLDEN, fname))
    X, _, nu = go.prologue(g["data"])
    out = ctx.gp_fit_batch(X[None], g["y_mean"][None], nu[None], int(g["n_iters"]))
    # Bounds per fixture (conftest.golden_tolerances): cov 1e-6, mu 1e-5, hypers 1e-5 for the nine members with R >= 6
    # realisations; the nine members with R <= 5 are chaotic over 2500 iterations (a one-ulp change of the target moves the
    # ORACLE's fit by up to 7e-5 on cov / 1.6e-3 on mu, tests/golden `chaos_*`), and a device that sums in another order is
    # such a change: they are held to ten times their measured envelope here - and TIGHTLY below, at a horizon where rounding
    # has not been amplified yet.  (The hypers are compared with the oracle's: the pickles do not hold them.)
    tol_cov, tol_mu, tol_hyp = golden_tolerances(g)
    e_cov, e_mu, e_hyp = relerr(out["cov"][0], g["cov_gold"]), np.max(np.abs(out["mu"][0] - g["mu_gold"])), relerr(out["hyp"][0], g["hyp"])
    print(f"{fname}: R={g['data'].shape[0]} cov {e_cov:.2e} (< {tol_cov:.1e}) mu {e_mu:.2e} (< {tol_mu:.1e}) hyp {e_hyp:.2e} (< {tol_hyp:.1e})")
    assert e_cov < tol_cov and e_mu < tol_mu and e_hyp < tol_hyp
    if float(g["chaos_cov"]) > 1e-12:
        # chaotic member: the same inputs for 100 iterations - one e-folding of the amplification - against the oracle
        short = ctx.gp_fit_batch(X[None], g["y_mean"][None], nu[None], 100)
        with threadpool_limits(1):
            mo, co, ho = go.gp_fit(X, g["y_mean"], nu, 100)
        s_cov, s_mu, s_hyp = relerr(short["cov"][0], co), relerr(short["mu"][0], mo), relerr(short["hyp"][0], ho)
        print(f"    T=100 vs oracle: cov {s_cov:.2e} mu {s_mu:.2e} hyp {s_hyp:.2e}")
        assert s_cov < 1e-11 and s_mu < 1e-11 and s_hyp < 1e-11          # (measured <= 5e-13 on all nine)
    # Distribution.__post_init__ (distrax): Cholesky + logdet of the reference's covariance
    chol, logdet = ctx.mvn_factor_batch(g["cov_gold"][None])
    assert np.max(np.abs(chol[0] - g["chol_gold"])) < 1e-11
    assert abs(logdet[0] - float(g["logdet_gold"])) < 1e-9



@pytest.mark.parametrize("fname", [f for f in golden_files() if "hist" in f])
def test_general_path_reproduces_the_references_pickled_posteriors_over_2500_iterations(ctx, fname):
    """The long horizon on the GENERAL route (csrc/driver.cpp on the batched primitives - the route of the N = 2000 headline): by
    size every fixture goes to a fused kernel, so the route is forced (bayens_set_fit_route) and the six N = 165 members of
    experiments/pre_fit_models/histssp*.pkl run their 2500 iterations (models.py:167,208-210) through it, to the same bars."""
    g = np.load(os.path.join(GOLDEN, fname))
    X, _, nu = go.prologue(g["data"])
    ctx.set_fit_route(1)
    try:
        out = ctx.gp_fit_batch(X[None], g["y_mean"][None], nu[None], int(g["n_iters"]))
        short = ctx.gp_fit_batch(X[None], g["y_mean"][None], nu[None], 100)
    finally:
        ctx.set_fit_route(0)
    fused = ctx.gp_fit_batch(X[None], g["y_mean"][None], nu[None], int(g["n_iters"]))
    tol_cov, tol_mu, tol_hyp = golden_tolerances(g)
    e_cov, e_mu, e_hyp = relerr(out["cov"][0], g["cov_gold"]), np.max(np.abs(out["mu"][0] - g["mu_gold"])), relerr(out["hyp"][0], g["hyp"])
    print(f"{fname} general route: cov {e_cov:.2e} (< {tol_cov:.1e}) mu {e_mu:.2e} (< {tol_mu:.1e}) hyp {e_hyp:.2e} (< {tol_hyp:.1e})")
    assert e_cov < tol_cov and e_mu < tol_mu and e_hyp < tol_hyp
    assert not np.array_equal(out["cov"], fused["cov"])                    # (two programs: else the route did not switch)
    with threadpool_limits(1):
        mo, co, ho = go.gp_fit(X, g["y_mean"], nu, 100)
    assert relerr(short["cov"][0], co) < 1e-10 and relerr(short["mu"][0], mo) < 1e-10 and relerr(short["hyp"][0], ho) < 1e-10


@pytest.mark.parametrize("N,R,T", [(320, 6, 500), (1000, 8, 100)])
def test_general_path_long_trajectories_stay_on_the_oracles(ctx, N, R, T):
    """N > 254 is the general route by size; the API default n_optim_nits = 500 (models.py:167) at N = 320 and 100 iterations at
    N = 1000, one member each against the oracle: mean, covariance, hyper-parameters and the loss samples at 1e-7."""
    data, _ = go.synthetic_ensemble(1, N, R=R, seed=N + T)
    X, y, nu = member_inputs(data)
    out = ctx.gp_fit_batch(X, y, nu, T, want_loss=True)
    assert not out["info"].any()
    with threadpool_limits(4):
        mo, co, ho, lo = go.gp_fit(X[0], y[0], nu[0], T, loss_every=25)
    e = dict(mu=relerr(out["mu"][0], mo), cov=relerr(out["cov"][0], co), hyp=relerr(out["hyp"][0], ho),
             loss=float(np.max(np.abs(out["loss"][0] - lo) / np.abs(lo))))
    print(f"general route N={N} T={T}: {e}")
    assert max(e.values()) < 1e-7


def test_fit_is_deterministic_and_independent_of_workspace_history(ctx):
    # regression: the Cholesky-backward solve once read 64-blocks of a reused workspace that nothing had written
    data, _ = go.synthetic_ensemble(2, 200, R=8, seed=21)
    X, y, nu = member_inputs(data)
    first = ctx.gp_fit_batch(X, y, nu, 3)
    other, _ = go.synthetic_ensemble(2, 200, R=8, seed=22)
    ctx.gp_fit_batch(*member_inputs(other), 2)              # dirties every workspace matrix
    again = ctx.gp_fit_batch(X, y, nu, 3)
    for k in ("mu", "cov", "chol", "logdet", "hyp", "basis"):
        assert np.array_equal(first[k], again[k]), k
    mo, co, ho = go.gp_fit(X[0], y[0], nu[0], 3)
    assert relerr(again["cov"][0], co) < 1e-10 and relerr(again["hyp"][0], ho) < 1e-12


def test_ragged_batch_members_do_not_interact(ctx):
    data, _ = go.synthetic_ensemble(4, 100, R=4, seed=5)
    X, y, nu = member_inputs(data)
    all_ = ctx.gp_fit_batch(X, y, nu, 2)
    one = ctx.gp_fit_batch(X[2:3], y[2:3], nu[2:3], 2)
    assert np.array_equal(all_["cov"][2], one["cov"][0]) and np.array_equal(all_["mu"][2], one["mu"][0])


def test_degenerate_member_is_reported_not_hidden(ctx):
    # a single realisation has zero variance across realisations: y_var = 0 -> the reference divides by zero too
    # (TF raises on the Cholesky); here the failing member is flagged through info / BAYENS_E_NOTPD.
    data, _ = go.synthetic_ensemble(2, 64, R=3, seed=1)
    X, y, nu = member_inputs(data)
    nu[1, :] = 0.0
    with pytest.raises(L.BayensError) as e:
        ctx.gp_fit_batch(X, y, nu, 1)
    assert e.value.code == -2 and e.value.info[1] > 0 and e.value.info[0] == 0


def test_bad_arguments_are_rejected(ctx):
    X = np.zeros((1, 8, 2))
    with pytest.raises(L.BayensError):
        ctx.gp_fit_batch(X, np.zeros((1, 8)), np.ones((1, 8)), 1, kernel=7)


def test_full_size_properties(ctx):
    """BASELINE config 2 shape (N=2000, R=8; 4 of the 32 members to bound the host-side check)."""
    M, N, R, T = 4, 2000, 8, 2
    data, obs = go.synthetic_ensemble(M, N, R=R)
    X, y, nu = member_inputs(data)
    out = ctx.gp_fit_batch(X, y, nu, T)
    assert not out["info"].any()
    for m in range(M):
        cov, ch = out["cov"][m], out["chol"][m]
        assert np.array_equal(cov, cov.T)
        assert relerr(ch @ ch.T, cov) < 1e-12                       # encode -> decode: chol(cov) chol(cov)^T == cov
        assert abs(np.sum(np.log(np.diag(ch))) - out["logdet"][m]) < 1e-8
        assert np.all(np.diag(cov) > nu[m])                          # cov = (PSD) + diag(y_var)
    # idempotence of the factorisation entry point on its own output
    ch2, ld2 = ctx.mvn_factor_batch(out["cov"][:1])
    assert relerr(ch2[0], out["chol"][0]) < 1e-13 and abs(ld2[0] - out["logdet"][0]) < 1e-9
    # one member against the oracle at full size
    mo, co, ho = go.gp_fit(X[0], y[0], nu[0], T)
    assert relerr(out["mu"][0], mo) < REL_TOL and relerr(out["cov"][0], co) < REL_TOL and relerr(out["hyp"][0], ho) < REL_TOL
    # (measured 1e-13: BASELINE's 1e-5 is the contract, this is the regression guard)
    assert relerr(out["mu"][0], mo) < 1e-9 and relerr(out["cov"][0], co) < 1e-9


@pytest.mark.parametrize("M,N,T", [(2, 86, 1), (3, 70, 27), (1, 128, 51), (2, 200, 26)])
def test_loss_trace_is_the_references_training_loss(ctx, M, N, T):
    """loss_trace = -ELBO sampled after iterations 0, 25, 50 (models.py:211-215), and asking for it changes nothing."""
    data, _ = go.synthetic_ensemble(M, N, R=4, seed=5)
    X, y, nu = member_inputs(data)
    out = ctx.gp_fit_batch(X, y, nu, T, want_loss=True)
    plain = ctx.gp_fit_batch(X, y, nu, T)
    assert plain["loss"] is None and out["loss"].shape == (M, (T + 24) // 25)
    for k in ("mu", "cov", "chol", "logdet", "hyp"):
        # bit for bit in the general path (the sample refactors C on the side).  The fused kernels (N <= 254) keep the first
        # iteration in its general form when the sample of iteration 0 needs C and S: the same numbers through another
        # (algebraically identical) route, equal to rounding
        assert np.array_equal(out[k], plain[k]) or (N <= 254 and T >= 2 and relerr(out[k], plain[k]) < 1e-11), k
    with threadpool_limits(1):
        for m in range(M):
            ref = go.gp_fit(X[m], y[m], nu[m], T, loss_every=25)[3]
            assert np.allclose(out["loss"][m], ref, rtol=1e-9, atol=0), (out["loss"][m], ref)
    ctx.set_groups(2)
    try:
        assert np.array_equal(ctx.gp_fit_batch(X, y, nu, T, want_loss=True)["loss"], out["loss"])
    finally:
        ctx.set_groups(0)


F32_TOL = 1e-3      # BASELINE.json: 1e-3 relative for fp32


@pytest.mark.parametrize("M,N,R,T,kernel", [
    (2, 86, 5, 2, 0), (3, 24, 3, 2, 0), (2, 200, 8, 2, 1),      # config 3 names Matern-5/2 (kernel id 1)
    (1, 64, 2, 1, 0), (2, 7, 3, 2, 0), (1, 130, 4, 27, 0), (1, 1000, 8, 2, 1),
    (2, 320, 5, 1, 0), (2, 320, 5, 3, 0), (1, 257, 4, 2, 1)])   # general path: one iteration (general form in fp32), several
def test_fp32_fit_against_the_fp64_oracle(ctx, M, N, R, T, kernel):
    """bayens_gp_fit_batch_f32 (fp32 storage and MFMA products) vs the float64 oracle: 1e-3 relative."""
    data, _ = go.synthetic_ensemble(M, N, R=R, seed=11)
    X, y, nu = member_inputs(data)
    out = ctx.gp_fit_batch(X, y, nu, T, kernel=kernel, dtype=np.float32, want_loss=True)
    assert not out["info"].any()
    assert out["mu"].dtype == np.float32 and out["cov"].dtype == np.float32 and out["hyp"].dtype == np.float64
    with threadpool_limits(1):
        for m in range(M):
            mo, co, ho, lo = go.gp_fit(X[m], y[m], nu[m], T, kernel=kernel, loss_every=25)
            Lc, ld = go.mvn_factor(co)
            assert relerr(out["mu"][m].astype(np.float64), mo) < F32_TOL
            assert relerr(out["cov"][m].astype(np.float64), co) < F32_TOL
            assert relerr(out["chol"][m].astype(np.float64), Lc) < F32_TOL
            assert abs(out["logdet"][m] - ld) < F32_TOL * max(1.0, abs(ld))
            assert relerr(out["hyp"][m], ho) < F32_TOL
            assert np.allclose(out["loss"][m], lo, rtol=F32_TOL)
            import scipy.linalg as sla
            assert relerr(out["basis"][m, 1].astype(np.float64), sla.solve_triangular(Lc, mo, lower=True)) < 10 * F32_TOL


def test_fp32_fuzz_case_whose_loss_sample_passes_near_zero(ctx):
    """Round 2's randomised sweep (profiles/r02_fuzz_seed11.log, case 12: M=3, N=65, R=8, T=3, Matern-5/2, fp32) reported
    "max rel err 2.88e-3" without saying of what.  It is the -ELBO SAMPLE of member 2: 0.0096, the sum of terms of size
    ~75 (sum_t 1/2 log nu_t = -75.3, the quadratic term +40.7, the KL +34.6).  An fp32 error of 3.7e-7 RELATIVE TO THE
    TERMS is 2.9e-3 relative to a loss that happens to pass near zero; the posterior itself (mu, cov, hypers) of the same
    fit sits at 1e-5.  The loss is therefore held to 1e-3 of the magnitude of its terms (which is what an fp32 evaluation
    can promise), everything else to BASELINE's 1e-3 - asserted 30 x tighter."""
    M, N, R, T, kernel = 3, 65, 8, 3, 1
    data, _ = go.synthetic_ensemble(M, N, R=R, seed=625353)
    X, y, nu = member_inputs(data)
    for dtype, tol in ((np.float32, F32_TOL), (np.float64, 1e-9)):
        out = ctx.gp_fit_batch(X, y, nu, T, kernel=kernel, dtype=dtype, want_loss=True)
        assert not out["info"].any()
        for m in range(M):
            mo, co, ho, lo = go.gp_fit(X[m], y[m], nu[m], T, kernel=kernel, loss_every=25)
            assert relerr(out["mu"][m].astype(np.float64), mo) < tol / 30 and relerr(out["hyp"][m], ho) < tol / 30
            assert relerr(out["cov"][m].astype(np.float64), co) < tol / 30
            scale = abs(np.sum(0.5 * go.LOG2PI + 0.5 * np.log(nu[m])))        # the loss's largest term
            assert scale > 60 and np.all(np.abs(out["loss"][m] - lo) < tol * scale), (m, out["loss"][m], lo)
    assert abs(lo[0]) < 0.02             # member 2: the sample that made the sweep's relative figure meaningless


def test_fp32_domain_is_the_conditioning_of_the_kernel_matrix(ctx):
    """Round 3's randomised sweep (profiles/r03_fuzz_seed47.log, case 37: M=5, N=400, R=3, T=30, Matern-3/2, fp32) found a member
    whose fp32 posterior mean is 3.8e-3 from the fp64 oracle.  It is the member, not the kernel: with three realisations one
    time step has a variance of 4e-7 across them and cond(K + J I) = 2.4e7 > 1 / eps_fp32 = 8e6, and a plain fp32 LAPACK
    restatement of the same arithmetic on the CPU misses 1e-3 on that member too (1.1e-3; the other four members: <= 9e-5 there,
    <= 1e-3 here).  The fp32 mode's 1e-3 is therefore a statement about members with cond(K + J I) well below 1 / eps_fp32 - the
    reference's Matern kernels over >= 4-5 realisations (config 3: R = 8) - and this test pins exactly that split."""
    M, N, R, T = 5, 400, 3, 30
    data, _ = go.synthetic_ensemble(M, N, R=R, seed=125637)
    X, y, nu = member_inputs(data)
    out = ctx.gp_fit_batch(X, y, nu, T, dtype=np.float32)
    assert not out["info"].any()
    worst_ok = 0.0
    with threadpool_limits(4):
        for m in range(M):
            mo, co, ho = go.gp_fit(X[m], y[m], nu[m], T)
            K0, _ = go.kernel_matrix(X[m], ho[0], ho[1])
            cond = np.linalg.cond(K0 + go.JITTER * np.eye(N))
            err = max(relerr(out["mu"][m].astype(np.float64), mo), relerr(out["cov"][m].astype(np.float64), co), relerr(out["hyp"][m], ho))
            if m == 3:
                assert cond > 2e7 and nu[m].min() < 1e-6 and err < 2e-2          # beyond fp32's domain: degraded, finite, flagged here
            else:
                worst_ok = max(worst_ok, err)
    assert worst_ok < F32_TOL


def test_fp32_full_size_is_finite_and_close_to_fp64(ctx):
    """Config 3 size (32-member batch trimmed to 4 for the test's run time): fp32 against the fp64 device path."""
    data, _ = go.synthetic_ensemble(4, 2000, R=8, seed=2)
    X, y, nu = member_inputs(data)
    o32 = ctx.gp_fit_batch(X, y, nu, 2, kernel=1, dtype=np.float32)
    o64 = ctx.gp_fit_batch(X, y, nu, 2, kernel=1)
    assert not o32["info"].any() and np.isfinite(o32["cov"]).all()
    assert relerr(o32["mu"].astype(np.float64), o64["mu"]) < F32_TOL
    assert relerr(o32["cov"].astype(np.float64), o64["cov"]) < F32_TOL
    assert relerr(o32["hyp"], o64["hyp"]) < F32_TOL


@pytest.mark.parametrize("M", [16, 32])
def test_benchmarked_batch_sizes_small_members_all_against_the_oracle(ctx, M):
    """The bench's batch (32 members) and config 4's per-GPU share (16): every member vs the oracle at N = 200."""
    data, _ = go.synthetic_ensemble(M, 200, R=8, seed=40 + M)
    X, y, nu = member_inputs(data)
    out = ctx.gp_fit_batch(X, y, nu, 2)
    assert not out["info"].any()
    with threadpool_limits(1):
        for m in range(M):
            mo, co, ho = go.gp_fit(X[m], y[m], nu[m], 2)
            Lc, ld = go.mvn_factor(co)
            assert relerr(out["mu"][m], mo) < 1e-10 and relerr(out["cov"][m], co) < 1e-10, m
            assert relerr(out["chol"][m], Lc) < 1e-10 and relerr(out["hyp"][m], ho) < 1e-10, m
            assert abs(out["logdet"][m] - ld) < 1e-9 * max(1.0, abs(ld)), m


@pytest.mark.parametrize("M", [16, 32])
def test_benchmarked_configuration_full_size(ctx, M):
    """BASELINE config 2 exactly as bench.py runs it (32 x 2000, R = 8, T = 2; 16 = config 4's per-GPU share), on
    resident buffers: members 0 / M/2-1 / M-1 against the oracle, EVERY member bit-equal to the same member fitted
    in a batch of 4 (the triangular GEMM launches deal tiles differently when nb % 8 == 0, gemm.hip), and the weighted
    posterior of the step (bayens_weighted_posterior: rows -> weights -> barycentre) against the oracle's
    weights.py:88-123 / ensemble_scheme.py:52-78 evaluated on the device's own (mu, chol, logdet)."""
    N, R, T = 2000, 8, 2
    data, obs = go.synthetic_ensemble(M, N, R=R, seed=0xB4E5)
    X, y, nu = member_inputs(data)
    dX, dy, dnu = ctx.to_device(X), ctx.to_device(y), ctx.to_device(nu)
    dmu, dcov, dchol = ctx.empty((M, N)), ctx.empty((M, N, N)), ctx.empty((M, N, N))
    dld, dhyp, dbasis = ctx.empty((M,)), ctx.empty((M, 2)), ctx.empty((M, 2, N))
    R_obs = 3
    dobs, dw, dbmu, dbvar = ctx.to_device(obs), ctx.empty((M, N)), ctx.empty((N,)), ctx.empty((N,))
    bufs = [dX, dy, dnu, dmu, dcov, dchol, dld, dhyp, dbasis, dobs, dw, dbmu, dbvar]
    try:
        info = ctx.gp_fit_batch_dev(M, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis)
        assert not info.any()
        # the weighted posterior exactly as bench.py's step runs it (one entry point on the resident posteriors)
        bad = ctx.weighted_posterior_dev(M, N, R_obs, dbasis, dld, dobs, dmu, dcov, dw, dbmu, dbvar, c_std=1e-3)
        w_dev, bmu_dev, bvar_dev = dw.download(), dbmu.download(), dbvar.download()
        mu, hyp, ld, basis = dmu.download(), dhyp.download(), dld.download(), dbasis.download()
        ll_ref, var_ref = np.empty((M, N)), np.empty((M, N))
        for b0 in range(0, M, 4):
            ref = ctx.gp_fit_batch(X[b0:b0 + 4], y[b0:b0 + 4], nu[b0:b0 + 4], T)
            assert np.array_equal(mu[b0:b0 + 4], ref["mu"]) and np.array_equal(hyp[b0:b0 + 4], ref["hyp"]), b0
            assert np.array_equal(ld[b0:b0 + 4], ref["logdet"]) and np.array_equal(basis[b0:b0 + 4], ref["basis"]), b0
            for j in range(4):     # covariance and factor member by member (32 MB each)
                cov = np.empty((N, N)); ch = np.empty((N, N))
                ctx._check(ctx.lib.bayens_memcpy_d2h(ctx.h, cov.ctypes.data, dcov.offset((b0 + j) * N * N * 8), cov.nbytes))
                ctx._check(ctx.lib.bayens_memcpy_d2h(ctx.h, ch.ctypes.data, dchol.offset((b0 + j) * N * N * 8), ch.nbytes))
                assert np.array_equal(cov, ref["cov"][j]) and np.array_equal(ch, ref["chol"][j]), b0 + j
                # oracle rows on the device's own (mu, chol, logdet): weights.py:98-104
                ll_ref[b0 + j] = go.loglik_mean_rows(go.mvn_constvec_loglik(mu[b0 + j], ch, ld[b0 + j], obs))
                var_ref[b0 + j] = np.diag(cov)
                if b0 + j in (0, M // 2 - 1, M - 1):
                    mo, co, ho = go.gp_fit(X[b0 + j], y[b0 + j], nu[b0 + j], T)
                    assert relerr(mu[b0 + j], mo) < 1e-9 and relerr(cov, co) < 1e-9 and relerr(hyp[b0 + j], ho) < 1e-9
        # north_star: "matching reference weights to 1e-5" at 32 x 2000 - the normalisation and the barycentre deal the
        # members over four waves (kernels.hip), so M = 32 is where that dealing is exercised at size
        w_ref, _ = go.loglik_weights(ll_ref, 1e-3)
        rm, rv, rbad = go.barycentre(w_ref, mu, var_ref)
        assert np.isfinite(w_ref).all() and bad == rbad == 0
        assert relerr(w_dev, w_ref) < REL_TOL and relerr(bmu_dev, rm) < REL_TOL and relerr(bvar_dev, rv) < REL_TOL
        assert relerr(w_dev, w_ref) < 1e-9 and relerr(bmu_dev, rm) < 1e-9 and relerr(bvar_dev, rv) < 1e-9     # (measured)
        # ... and element by element: every weight above underflow to 1e-5 RELATIVE to itself (norm-wise bounds say nothing
        # about the small entries of a weight table)
        big = w_ref > 1e-300
        assert big.all() and np.max(np.abs(w_dev[big] / w_ref[big] - 1.0)) < REL_TOL
        assert np.max(np.abs(w_dev[big] / w_ref[big] - 1.0)) < 1e-8                                             # (measured)
        assert np.allclose(w_dev.sum(axis=0), 1.0, rtol=1e-12)
    finally:
        for b in bufs:
            b.free()


def test_rbf_bench_leg_full_size_against_the_oracle(ctx):
    """bench.py's `rbf_f64` leg (BASELINE.json configs[1] as worded: "32-member x 2000-timestep synthetic, RBF GP fp64") exactly as
    bench.py runs it - Workload(ctx, 32, 2000, 8, T = 2, "rbf"): bench's seed, resident buffers, one batched fit - with members 0
    and 31 against the oracle at BASELINE's 1e-5.  RBF has no reference call site (SURVEY App. A "extension kernels"): the pin is
    the oracle's restatement of GPflow 2.1.5's SquaredExponential."""
    M, N, R, T = 32, 2000, 8, 2
    data, obs = go.synthetic_ensemble(M, N, R=R, seed=0xB4E5)
    X, y, nu = member_inputs(data)
    dX, dy, dnu = ctx.to_device(X), ctx.to_device(y), ctx.to_device(nu)
    dmu, dcov, dchol = ctx.empty((M, N)), ctx.empty((M, N, N)), ctx.empty((M, N, N))
    dld, dhyp, dbasis = ctx.empty((M,)), ctx.empty((M, 2)), ctx.empty((M, 2, N))
    bufs = [dX, dy, dnu, dmu, dcov, dchol, dld, dhyp, dbasis]
    try:
        info = ctx.gp_fit_batch_dev(M, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, basis=dbasis, kernel=go.KERNEL_RBF)
        assert not info.any()
        mu, hyp, ld = dmu.download(), dhyp.download(), dld.download()
        assert np.isfinite(mu).all() and np.isfinite(ld).all()
        for m in (0, M - 1):
            cov = np.empty((N, N)); ch = np.empty((N, N))
            ctx._check(ctx.lib.bayens_memcpy_d2h(ctx.h, cov.ctypes.data, dcov.offset(m * N * N * 8), cov.nbytes))
            ctx._check(ctx.lib.bayens_memcpy_d2h(ctx.h, ch.ctypes.data, dchol.offset(m * N * N * 8), ch.nbytes))
            mo, co, ho = go.gp_fit(X[m], y[m], nu[m], T, kernel=go.KERNEL_RBF)
            Lc, ldo = go.mvn_factor(co)
            assert relerr(mu[m], mo) < REL_TOL and relerr(cov, co) < REL_TOL and relerr(hyp[m], ho) < REL_TOL, m
            assert relerr(ch, Lc) < REL_TOL and abs(ld[m] - ldo) < REL_TOL * abs(ldo), m
            assert np.array_equal(cov, cov.T)
    finally:
        for b in bufs:
            b.free()


def test_ill_conditioned_members_against_extended_precision(ctx):
    """Members with a variance of ~1e-11 across realisations at one time step (two realisations 6e-6 apart): cond(Lam) ~ 1e11, so
    an fp64 evaluation of the fit is determined to cond * eps ~ 1e-5 at best - and "device vs oracle" then compares two fp64
    answers.  The arbiter is the 50-digit twin of the oracle (oracle/gp_oracle_mp.py -> tests/golden/illcond_*.npz, which also
    records that the fp64 oracle itself is <= 1e-6 from it).  Round 3's sweep (profiles/r03_fuzz_seed32.log, case 21 = illcond_s32c21)
    had the fused kernels 2.3 (relative) off on the mean: m = S theta1 through the explicit S, then u = G m - L^T (y / nu).  Every
    route is held to BASELINE's 1e-5 against the 50-digit answer: the fused small-member kernel (N = 24), the general path
    (the same member, bayens_set_fit_route), the fused mid-size kernel (N = 96), with and without loss samples (the sample of
    iteration 0 keeps the first iteration in its general form)."""
    for tag, routes in (("illcond_s32c21", (0, 1)), ("illcond_n96", (0, 1))):
        g = np.load(os.path.join(GOLDEN, tag + ".npz"))
        X, y, nu = go.prologue(g["data"])
        assert nu.min() < 1e-9 * nu.max()
        for T in g["T"]:
            T = int(T)
            assert np.all(g[f"oracle_err_T{T}"] < 1e-5) and g[f"cond_lam_T{T}"] > 1e10
            for route in routes:
                for want_loss in (False, True):
                    ctx.set_fit_route(route)
                    try:
                        out = ctx.gp_fit_batch(X[None], y[None], nu[None], T, want_loss=want_loss)
                    finally:
                        ctx.set_fit_route(0)
                    where = (tag, T, "general" if route else "fused", want_loss)
                    assert not out["info"].any(), where
                    assert relerr(out["mu"][0], g[f"mu_T{T}"]) < REL_TOL, (where, relerr(out["mu"][0], g[f"mu_T{T}"]))
                    assert relerr(out["cov"][0], g[f"cov_T{T}"]) < REL_TOL, (where, relerr(out["cov"][0], g[f"cov_T{T}"]))
                    assert relerr(out["hyp"][0], g[f"hyp_T{T}"]) < 1e-8, (where, relerr(out["hyp"][0], g[f"hyp_T{T}"]))


def test_config3_full_size_against_the_fp64_oracle(ctx):
    """BASELINE config 3 at size: fp32 Matern-5/2, N = 2000 - one member against the float64 ORACLE (1e-3)."""
    data, _ = go.synthetic_ensemble(2, 2000, R=8, seed=2)
    X, y, nu = member_inputs(data)
    o32 = ctx.gp_fit_batch(X, y, nu, 2, kernel=1, dtype=np.float32)
    assert not o32["info"].any()
    mo, co, ho = go.gp_fit(X[0], y[0], nu[0], 2, kernel=1)
    Lc, ld = go.mvn_factor(co)
    assert relerr(o32["mu"][0].astype(np.float64), mo) < F32_TOL and relerr(o32["cov"][0].astype(np.float64), co) < F32_TOL
    assert relerr(o32["chol"][0].astype(np.float64), Lc) < F32_TOL and relerr(o32["hyp"][0], ho) < F32_TOL
    assert abs(o32["logdet"][0] - ld) < F32_TOL * abs(ld)


def test_member_groups_do_not_change_the_result(ctx):
    """The general route can split a batch over member groups = streams (include/bayens.h bayens_set_groups; one group by
    default): one, two, three groups and the default give the same bits (N = 130 is forced onto the general route; N = 320 with 32
    members takes it by size)."""
    data, _ = go.synthetic_ensemble(5, 130, R=4, seed=31)
    X, y, nu = member_inputs(data)
    ctx.set_fit_route(1)
    try:
        ctx.set_groups(1)
        one = ctx.gp_fit_batch(X, y, nu, 2)
        ctx.set_groups(3)
        three = ctx.gp_fit_batch(X, y, nu, 2)
    finally:
        ctx.set_groups(0)
        ctx.set_fit_route(0)
    for k in ("mu", "cov", "chol", "logdet", "hyp", "basis"):
        assert np.array_equal(one[k], three[k]), k
    data, _ = go.synthetic_ensemble(32, 320, R=3, seed=32)
    X, y, nu = member_inputs(data)
    auto = ctx.gp_fit_batch(X, y, nu, 2, want_chol=False)
    ctx.set_groups(2)
    try:
        halves = ctx.gp_fit_batch(X, y, nu, 2, want_chol=False)
    finally:
        ctx.set_groups(0)
    for k in ("mu", "cov", "hyp"):
        assert np.array_equal(auto[k], halves[k]), k


def test_bench_contract_line(ctx):
    """bench.py prints ONE JSON line with the driver's contract fields (small shape so it runs in seconds)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--members", "4", "--timesteps", "200",
                          "--steps", "2", "--warmup", "1", "--cpu-members", "1"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["value"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["peak"] == 78.6
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0


@pytest.mark.parametrize("M,N,R,T", [(2, 1, 3, 1), (2, 2, 3, 2), (1, 3, 2, 2), (2, 63, 3, 2), (2, 64, 3, 2), (2, 65, 3, 2),
                                     (1, 127, 50, 1), (1, 129, 4, 2), (1, 255, 6, 1), (1, 257, 6, 1), (3, 320, 7, 1),
                                     (1, 1000, 8, 1), (1, 1980, 8, 1), (33, 100, 4, 1),
                                     # T >= 2 in the general path: the symmetric first iteration with its paired factorisation
                                     # (T = 1 above keeps C and takes G S = (I - S) / gamma), with and without padding
                                     (1, 257, 6, 2), (3, 320, 7, 3), (5, 448, 4, 2), (1, 1980, 8, 2)])
def test_edge_sizes(ctx, M, N, R, T):
    """N = 1, around the 64-row leaf, the 256-wide panel and the 128 tile; the monthly horizon 1980; M not a
    multiple of 8; many realisations."""
    data, obs = go.synthetic_ensemble(M, N, R=R, seed=N * 7 + R)
    X, y, nu = member_inputs(data)
    out = ctx.gp_fit_batch(X, y, nu, T)
    assert not out["info"].any()
    for m in range(min(M, 2)):
        mo, co, ho = go.gp_fit(X[m], y[m], nu[m], T)
        Lc, ld = go.mvn_factor(co)
        assert relerr(out["mu"][m], mo) < 1e-10 and relerr(out["cov"][m], co) < 1e-10
        assert relerr(out["chol"][m], Lc) < 1e-10 and relerr(out["hyp"][m], ho) < 1e-10
        assert abs(out["logdet"][m] - ld) < 1e-9 * max(1.0, abs(ld))
    ll = ctx.loglik_rows_basis(out["basis"], out["logdet"], obs)
    assert relerr(ll, ctx.loglik_rows(out["mu"], out["chol"], out["logdet"], obs)) < 1e-12
