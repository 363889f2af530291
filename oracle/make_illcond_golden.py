"""Writes tests/golden/illcond_*.npz (TEST INFRASTRUCTURE ONLY): members whose variance across realisations all but
vanishes at one time step, with the fit's answer from the 50-digit twin of the oracle (oracle/gp_oracle_mp.py).

Why: round 3's randomised sweep (profiles/r03_fuzz_seed32.log, case 21: N = 24, R = 2, T = 5, Matern-3/2, fp64) found the
device 2.3 (relative) from the fp64 oracle on the posterior mean of a member with min(nu) = 9.6e-12, cond(Lam) ~ 1e11 - and
took the case out of the count instead of asking which of the two was right.  With the multi-precision answer as the
arbiter: the oracle is 5e-7 from it (cond * eps = 1e-5 is the honest bound), the device was not - it formed m = S theta1
through the EXPLICIT S = C^-T C^-1 (error eps * |theta1| ~ 1e-16 * 1e11 in m) and then u = G m - L^T (y / nu), which
multiplies that error by |G| ~ 1e11.  tests/test_gpu_fit_parity.py::test_ill_conditioned_members_against_extended_precision
holds every route of the device to these fixtures.

  illcond_s32c21   the sweep's own case (seed 32, case 21; data seed 135174): N = 24, R = 2 -> fused small-member kernel
  illcond_n96      the same construction at N = 96, R = 2 (one time step's two realisations 6e-6 apart) -> fused mid-size kernel
(the general path is held to illcond_s32c21 too: the test forces it with bayens_set_fit_route)

Usage: python oracle/make_illcond_golden.py [tag ...]
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gp_oracle as go  # noqa: E402
from oracle import gp_oracle_mp as gm  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def case_inputs(tag):
    if tag == "illcond_s32c21":
        data, _ = go.synthetic_ensemble(1, 24, R=2, seed=135174)
        return data[0], [1, 2, 3, 5]
    if tag == "illcond_n96":
        data, _ = go.synthetic_ensemble(1, 96, R=2, seed=96)
        data[0, 1, 71] = data[0, 0, 71] + 6e-6             # nu = 9e-12 at one time step
        return data[0], [1, 3]
    raise KeyError(tag)


def main():
    tags = sys.argv[1:] or ["illcond_s32c21", "illcond_n96"]
    rel = lambda a, b: float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
    for tag in tags:
        data, Ts = case_inputs(tag)
        X, y, nu = go.prologue(data)
        out = dict(data=data, T=np.array(Ts))
        print(f"[{tag}] N={X.shape[0]} R={X.shape[1]} min(nu)={nu.min():.3e} max(nu)={nu.max():.3e}", flush=True)
        for T in Ts:
            t0 = time.time()
            mu, cov, hyp, _ = gm.gp_fit_mp(X, y, nu, T)
            mu, cov, hyp = gm.to_f64(mu), gm.to_f64(cov), gm.to_f64(hyp)
            mo, co, ho, st = go.gp_fit(X, y, nu, T, return_state=True)
            out[f"mu_T{T}"], out[f"cov_T{T}"], out[f"hyp_T{T}"] = mu, cov, hyp
            out[f"oracle_err_T{T}"] = np.array([rel(mo, mu), rel(co, cov), rel(ho, hyp)])
            out[f"cond_lam_T{T}"] = np.float64(np.linalg.cond(st.Lam))
            print(f"   T={T}: fp64 oracle vs 50 digits: mu {rel(mo, mu):.2e} cov {rel(co, cov):.2e} hyp {rel(ho, hyp):.2e}; "
                  f"cond(Lam) {np.linalg.cond(st.Lam):.2e}  ({time.time() - t0:.0f} s)", flush=True)
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)


if __name__ == "__main__":
    main()
