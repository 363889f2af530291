"""Randomised sweep of the device fit against the oracle: shapes, iteration counts, kernels, precisions, loss traces."""
import os, sys
import numpy as np
from threadpoolctl import threadpool_limits
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesian_ensembling_amd import _lib as L
from oracle import gp_oracle as go
ctx = L.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
rel = lambda a, b: float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
worst = {"f64": 0.0, "f32": 0.0}
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
with threadpool_limits(4):
    for case in range(n):
        M = int(rng.integers(1, 6)); N = int(rng.choice([1, 2, 5, 24, 63, 64, 65, 86, 127, 128, 129, 165, 200, 255, 257, 300, 400, 513]))
        R = int(rng.integers(1, 12)); T = int(rng.choice([0, 1, 2, 3, 5, 26, 30])); kern = int(rng.integers(0, 3))
        if kern != 0 and N > 300:
            kern = 0                      # smoother kernels at larger N: K + 1e-6 I leaves fp64 itself at 1e-6
        # fp32 is for well-conditioned kernel matrices: the reference's Matern kernels over >= 3 realisations (1-D
        # inputs or the RBF kernel with 1e-6 jitter lose positive definiteness in fp32 - reported through info[])
        f32 = bool(rng.integers(0, 2)) and N >= 5 and R >= 3 and kern != 2
        data, _ = go.synthetic_ensemble(M, N, R=R, seed=int(rng.integers(0, 10**6)))
        X = np.ascontiguousarray(np.swapaxes(data, 1, 2)); y = data.mean(axis=1); nu = data.var(axis=1) + (1e-3 if R == 1 else 0.0)
        try:
            out = ctx.gp_fit_batch(X, y, nu, T, kernel=kern, dtype=np.float32 if f32 else np.float64, want_loss=True)
        except L.BayensError as ex:
            note = "   <-- CHECK"
            info = getattr(ex, "info", None)
            if f32 and info is not None and np.any(info):
                # a failed pivot in fp32: is the member inside the mode's stated domain at all?  cond(K + J I) at the INITIAL hypers
                # (the fit did not get further) against 1 / eps_fp32 = 8e6
                conds = []
                for m in np.nonzero(info)[0]:
                    K0, _ = go.kernel_matrix(X[m], 1.0, 1.0, kernel=kern)
                    ev = np.linalg.eigvalsh(K0 + go.JITTER * np.eye(N))
                    conds.append(float(ev[-1] / max(ev[0], 1e-300)))
                if min(conds) > 1e6:
                    note = f"   (outside fp32's domain: cond(K + J I) = {min(conds):.1e} at the initial hypers; reported through info[], as designed)"
                    worst["f32_refused_outside_domain"] = worst.get("f32_refused_outside_domain", 0) + 1
            print(f"case {case:3d} M={M} N={N:4d} R={R:2d} T={T:2d} k={kern} {'f32' if f32 else 'f64'}: {ex} info {info}{note}", flush=True)
            continue
        errs = {"mu": 0.0, "cov": 0.0, "hyp": 0.0, "loss": 0.0}
        cond_k = 0.0                      # fp32 cases: cond(K + J I) at the fit's final hypers (the fp32 mode's stated domain)
        for m in range(M):
            res = go.gp_fit(X[m], y[m], nu[m], T, kernel=kern, loss_every=25) if T > 0 else go.gp_fit(X[m], y[m], nu[m], T, kernel=kern)
            mo, co, ho = res[0], res[1], res[2]
            if f32:
                K0, _ = go.kernel_matrix(X[m], ho[0], ho[1], kernel=kern)
                ev = np.linalg.eigvalsh(K0 + go.JITTER * np.eye(N))
                cond_k = max(cond_k, float(ev[-1] / max(ev[0], 1e-300)))
            if np.max(np.abs(mo)) > 1e-12:
                errs["mu"] = max(errs["mu"], rel(out["mu"][m].astype(float), mo))
            errs["cov"] = max(errs["cov"], rel(out["cov"][m].astype(float), co))
            errs["hyp"] = max(errs["hyp"], rel(out["hyp"][m], ho))
            if T > 0:
                # the -ELBO is a cancelling sum (sum of 1/2 log nu against the quadratic and KL terms): its error is measured
                # against the magnitude of its largest term, not against a value that may pass near zero (round 2's "2.9e-3")
                scale = max(abs(float(np.sum(0.5 * go.LOG2PI + 0.5 * np.log(nu[m])))), float(np.max(np.abs(res[3]))), 1e-300)
                errs["loss"] = max(errs["loss"], float(np.max(np.abs(out["loss"][m] - res[3])) / scale))
        e = max(errs.values())
        key = "f32" if f32 else "f64"
        # two (nearly) coinciding realisations leave a variance of ~0 at a time point: cond(Lam) ~ max(nu) / min(nu) and an fp64
        # fit is determined to cond * eps at best (seed 32, case 21: nu_min 1e-11 - the 50-digit answer of oracle/gp_oracle_mp.py
        # puts the oracle 5e-7 and, since round 4, the device 5e-6 from the truth; rounds 2-3 had the device O(1) off: m = S theta1
        # through the explicit S).  Such cases count against cond * eps, in a column of their own.
        ratio = float(nu.max()) / max(float(nu.min()), 1e-300)
        degenerate = ratio > 1e9
        # fp32's stated domain: members with cond(K + J I) well below 1 / eps_fp32 = 8e6.  Beyond 1e6 a case counts
        # in a column of its own against a loose bound (degraded, finite, reported) - the reference's Matern kernels over >= 4-5
        # realisations stay inside; R = 3 with a near-coincident pair of realisations does not
        outside = f32 and cond_k > 1e6
        bound = (1e-3 if f32 else 1e-6) if not degenerate else min(1e-2, 100 * ratio * 2.2e-16)
        if outside:
            bound = 2e-1
        col = key + ("_illcond" if degenerate else ("_cond_above_1e6" if outside else ""))
        flag = "" if e < bound else "   <-- CHECK"
        if e >= bound and not f32 and not degenerate and T > 0:
            # Before a case counts as a defect: how far does the ORACLE's own answer move when its target moves by one ulp?  (Members
            # of two or three realisations amplify rounding over the iterations - the reference's R <= 5 fixtures do the same,
            # DESIGN section 2 - and a device that sums in another order lands somewhere inside that envelope.)
            env = 0.0
            for m in range(M):
                base = go.gp_fit(X[m], y[m], nu[m], T, kernel=kern, loss_every=25)
                lscale = max(abs(float(np.sum(0.5 * go.LOG2PI + 0.5 * np.log(nu[m])))), float(np.max(np.abs(base[3]))), 1e-300)
                for trial in range(8):                      # (the envelope is heavy-tailed: eight draws, mean / covariance / hypers / loss)
                    yp = y[m] * (1.0 + 2.2e-16 * np.random.default_rng(trial).choice([-1.0, 0.0, 1.0], size=N))
                    pert = go.gp_fit(X[m], yp, nu[m], T, kernel=kern, loss_every=25)
                    env = max(env, rel(pert[2], base[2]), rel(pert[1], base[1]), rel(pert[0], base[0]) if np.max(np.abs(base[0])) > 1e-12 else 0.0,
                              float(np.max(np.abs(pert[3] - base[3])) / lscale))
            if e < 10.0 * env:
                flag = f"   (chaotic member: the oracle's own answer moves {env:.1e} under a one-ulp change of its target)"
                col = key + "_chaotic"
        worst[col] = max(worst.get(col, 0.0), e)
        if outside:
            flag += f"   (outside fp32's domain: cond(K + J I) = {cond_k:.1e})"
        if degenerate:
            flag += f"   (ill-conditioned: max(nu)/min(nu) = {ratio:.1e}, bound {bound:.1e})"
        which = max(errs, key=errs.get)
        print(f"case {case:3d} M={M} N={N:4d} R={R:2d} T={T:2d} k={kern} {key}: max rel err {e:.2e} ({which}) info {int(out['info'].max())}{flag}", flush=True)
print("worst", worst)
