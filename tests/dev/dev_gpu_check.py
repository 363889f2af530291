"""Developer check run on the GPU box: primitives vs NumPy, fit vs oracle, first timings."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesian_ensembling_amd import _lib as L  # noqa: E402
from oracle import gp_oracle as go  # noqa: E402

ctx = L.Context(0)
print(ctx.device_info(), flush=True)
rng = np.random.default_rng(0)


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


# ---- gemm ----
for (nb, m, n, k) in [(2, 128, 128, 64), (1, 192, 64, 128), (3, 320, 256, 192), (1, 64, 64, 64)]:
    for ta in (0, 1):
        for tb in (0, 1):
            A = rng.standard_normal((nb, k, m) if ta else (nb, m, k))
            B = rng.standard_normal((nb, n, k) if tb else (nb, k, n))
            C0 = rng.standard_normal((nb, m, n))
            flags = (L.GEMM_TA if ta else 0) | (L.GEMM_TB if tb else 0)
            out = ctx.test_gemm(A, B, C0, alpha=0.7, beta=-0.3, flags=flags, diag_add=0.25)
            opA = np.swapaxes(A, 1, 2) if ta else A
            opB = np.swapaxes(B, 1, 2) if tb else B
            ref = 0.7 * opA @ opB - 0.3 * C0
            idx = np.arange(min(m, n))
            ref[:, idx, idx] += 0.25
            print("gemm", nb, m, n, k, ta, tb, "relerr %.2e" % relerr(out, ref), flush=True)

# tri_out + ktrim (TN, lower operands)
nb, n = 2, 256
A = np.tril(rng.standard_normal((nb, n, n)))
B = np.tril(rng.standard_normal((nb, n, n)))
C0 = rng.standard_normal((nb, n, n))
out = ctx.test_gemm(A, B, C0, alpha=1.0, beta=0.0,
                    flags=L.GEMM_TA | L.GEMM_TRI_OUT | L.GEMM_KTRIM_TN_LOWER)
phi = np.swapaxes(A, 1, 2) @ B
blk = (np.arange(n)[:, None] // 64) >= (np.arange(n)[None, :] // 64)
print("gemm tri/ktrim relerr %.2e ; untouched upper blocks kept: %s" % (
    relerr(out[:, blk], phi[:, blk]), np.array_equal(out[:, ~blk], C0[:, ~blk])), flush=True)

# ---- potrf ----
for (nb, n) in [(2, 64), (2, 192), (1, 512)]:
    M = rng.standard_normal((nb, n, n))
    S = M @ np.swapaxes(M, 1, 2) + n * np.eye(n)
    Lg, info = ctx.test_potrf(S)
    ref = np.linalg.cholesky(S)
    print("potrf", nb, n, "relerr %.2e" % relerr(Lg, ref), info, flush=True)

# ---- trsm ----
for (nb, m, n) in [(2, 128, 192), (1, 256, 256)]:
    M = rng.standard_normal((nb, n, n))
    Lt = np.linalg.cholesky(M @ np.swapaxes(M, 1, 2) + n * np.eye(n))
    Bm = rng.standard_normal((nb, m, n))
    X0 = ctx.test_trsm(Lt, Bm, 0)
    X1 = ctx.test_trsm(Lt, Bm, 1)
    r0 = np.stack([np.linalg.solve(Lt[b], Bm[b].T).T for b in range(nb)])          # B L^-T
    r1 = np.stack([np.linalg.solve(Lt[b].T, Bm[b].T).T for b in range(nb)])        # B L^-1
    print("trsm", nb, m, n, "rlt %.2e rln %.2e" % (relerr(X0, r0), relerr(X1, r1)), flush=True)
nb, n = 1, 256
M = rng.standard_normal((nb, n, n))
Lt = np.linalg.cholesky(M @ np.swapaxes(M, 1, 2) + n * np.eye(n))
Bm = np.tril(rng.standard_normal((nb, n, n)))
X2 = ctx.test_trsm(Lt, Bm, 2)
r2 = np.stack([np.linalg.solve(Lt[b].T, Bm[b].T).T for b in range(nb)])
print("trsm rln lower-rhs %.2e" % relerr(np.tril(X2), np.tril(r2)), flush=True)

# ---- fit vs oracle ----
for (M_, N, R, T, kern) in [(2, 86, 5, 3, 0), (3, 200, 8, 2, 0), (1, 130, 4, 2, 1), (1, 100, 8, 2, 2)]:
    data, obs = go.synthetic_ensemble(M_, N, R=R)
    Xs, ys, nus = zip(*[go.prologue(data[m]) for m in range(M_)])
    X, y, nu = np.stack(Xs), np.stack(ys), np.stack(nus)
    t0 = time.time()
    out = ctx.gp_fit_batch(X, y, nu, T, kernel=kern)
    dt = time.time() - t0
    for m in range(M_):
        mo, co, ho = go.gp_fit(X[m], y[m], nu[m], T, kernel=kern)
        Lc, ld = go.mvn_factor(co)
        print("fit N=%d R=%d T=%d k=%d: mu %.2e cov %.2e chol %.2e logdet %.2e hyp %.2e info %d (%.3fs)" % (
            N, R, T, kern, relerr(out["mu"][m], mo), relerr(out["cov"][m], co), relerr(out["chol"][m], Lc),
            abs(out["logdet"][m] - ld), relerr(out["hyp"][m], ho), out["info"][m], dt), flush=True)
    # weights path
    ll = ctx.loglik_rows(out["mu"], out["chol"], out["logdet"], obs)
    ll_ref = np.stack([go.loglik_mean_rows(go.mvn_constvec_loglik(out["mu"][m], out["chol"][m], out["logdet"][m], obs))
                       for m in range(M_)])
    w = ctx.weights_normalise(ll, 1e-3)
    w_ref, _ = go.loglik_weights(ll_ref, 1e-3)
    var = ctx.extract_diag(out["cov"])
    bm, bv, bad = ctx.barycentre(w, out["mu"], var)
    rm, rv, rbad = go.barycentre(w_ref, out["mu"], var)
    wt = ctx.w2_temporal(out["mu"], var)
    wt_ref = go.w2_weights_temporal(out["mu"], var)
    print("   ll %.2e w %.2e bary mu %.2e var %.2e (bad %d/%d) w2t %.2e" % (
        relerr(ll, ll_ref), relerr(w, w_ref), relerr(bm, rm), relerr(bv, rv), bad, rbad, relerr(wt, wt_ref)), flush=True)

# ---- first timing at the headline size ----
if "--time" in sys.argv:
    for (M_, N, R, T) in [(8, 2000, 8, 2), (32, 2000, 8, 2)]:
        data, obs = go.synthetic_ensemble(M_, N, R=R)
        X = np.ascontiguousarray(np.swapaxes(data, 1, 2))
        y = data.mean(axis=1)
        nu = data.var(axis=1)
        dX, dy, dnu = ctx.to_device(X), ctx.to_device(y), ctx.to_device(nu)
        dmu, dcov, dchol = ctx.empty((M_, N)), ctx.empty((M_, N, N)), ctx.empty((M_, N, N))
        dld, dhyp = ctx.empty((M_,)), ctx.empty((M_, 2))
        for rep in range(3):
            ctx.timer_start()
            info = ctx.gp_fit_batch_dev(M_, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp)
            ms = ctx.timer_stop()
            fl = M_ * (T * (13 / 3 * N**3 + 2 * N * N * R) + 13 / 3 * N**3 + 2 * N * N * R + N**3 / 3)
            print("time M=%d N=%d T=%d: %.1f ms  -> %.1f fits/s, %.2f TFLOP/s algorithmic; info max %d" % (
                M_, N, T, ms, M_ / (ms * 1e-3), fl / (ms * 1e-3) / 1e12, int(info.max())), flush=True)
        ctx.profile_enable(True)
        ctx.gp_fit_batch_dev(M_, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp)
        p = ctx.profile_get(0)
        ctx.profile_enable(False)
        print("   gemm launches %d, %.1f ms, %.2f TFLOP/s" % (p["launches"], p["ms"], p["flop"] / (p["ms"] * 1e-3) / 1e12), flush=True)
        hyp = dhyp.download()
        mo, co, ho = go.gp_fit(X[0], y[0], nu[0], T)
        print("   member0 vs oracle: mu %.2e cov %.2e hyp %.2e" % (
            relerr(dmu.download()[0], mo), relerr(dcov.download((1, N, N))[0], co), relerr(hyp[0], ho)), flush=True)
        for d in (dX, dy, dnu, dmu, dcov, dchol, dld, dhyp):
            d.free()
print("DONE", flush=True)
