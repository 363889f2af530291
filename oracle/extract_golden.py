"""Build tests/golden/*.npz from the reference's pickled fits (TEST INFRASTRUCTURE ONLY).

Runs in the build container only (needs /root/reference).  It never imports reference or third-party
code: the pickles are read with a restricted Unpickler whose find_class returns inert stub classes for
everything except numpy's array reconstruction helpers.

Per selected member the fixture holds
    data (R,N)            model_data._variable._data                      input of GPDTW1D.fit
    mu_gold (N,)          _distribution.mu                                 output of GPDTW1D.fit (T=2500)
    cov_gold (N,N)        _distribution.covariance
    chol_gold (N,N)       _distribution._dist._scale_tri                   distrax cached Cholesky
    logdet_gold ()        TriangularLinear._logdet
    y_mean (N,)           the DTW-barycentre target the reference used (tslearn, absent here), *recovered*
                          from mu_gold by the fixed point  y <- y + T^-1 (mu_gold - mu(y))  of SURVEY App. B
    hyp (2,)              (variance, lengthscale) the oracle reaches with that y_mean after T=2500
    n_iters               2500 (experiments/pre_fitting_cmip6models.py:76-77)

Usage:  python oracle/extract_golden.py [--only TAG] [--missing] [--jobs 6]
"""
from __future__ import annotations

import argparse
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gp_oracle as go  # noqa: E402

REF = "/root/reference/experiments/pre_fit_models"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# (tag, pickle, member index): ALL 18 pickled members (round 4; rounds 1-3 carried four of them).
#   histssp434 / histssp460: N=165, R in {25, 9, 16} / {25, 9, 10};  ssp119: N=86, R in {25, 25, 5, 6, 3, 5};
#   ssp434: N=86, R = 5, 5, 5;  ssp460: N=86, R in {5, 5, 4}
MEMBERS_PER_PICKLE = {"histssp434": 3, "histssp460": 3, "ssp119": 6, "ssp434": 3, "ssp460": 3}
SELECTION = [(f"{stem}_m{i}", f"{stem}_1D_models.pkl", i) for stem, n in MEMBERS_PER_PICKLE.items() for i in range(n)]


class _Stub:
    def __init__(self, *a, **k):
        self._args = a
        self._kw = k

    def __setstate__(self, s):
        self._state = s

    def __call__(self, *a, **k):
        return _Stub(*a, **k)


_ALLOW = {
    ("numpy.core.multiarray", "_reconstruct"),
    ("numpy", "ndarray"),
    ("numpy", "dtype"),
    ("numpy.core.multiarray", "scalar"),
    ("builtins", "str"),
}


class _Unpickler(pickle.Unpickler):
    def find_class(self, mod, name):
        if (mod, name) in _ALLOW:
            return super().find_class(mod, name)
        return type(name, (_Stub,), {"_mod": mod})


def _state(o):
    s = getattr(o, "_state", None)
    if isinstance(s, tuple):
        d = {}
        for p in s:
            if isinstance(p, dict):
                d.update(p)
        return d
    return s


def _find_logdet(obj, depth=0):
    """Depth-first search of the stub tree for the TriangularLinear '_logdet' scalar."""
    if depth > 8:
        return None
    s = _state(obj) if isinstance(obj, _Stub) else obj
    if isinstance(s, dict):
        if "_logdet" in s:
            return s["_logdet"]
        for v in s.values():
            r = _find_logdet(v, depth + 1)
            if r is not None:
                return r
    elif isinstance(s, (list, tuple)):
        for v in s:
            r = _find_logdet(v, depth + 1)
            if r is not None:
                return r
    if isinstance(obj, _Stub):
        for v in list(getattr(obj, "_args", ())) + list(getattr(obj, "_kw", {}).values()):
            r = _find_logdet(v, depth + 1)
            if r is not None:
                return r
    return None


def load_members(path):
    with open(path, "rb") as f:
        mc = _Unpickler(f).load()
    out = []
    for pm in _state(mc)["models"]:
        s = _state(pm)
        var = _state(_state(s["model_data"])["_variable"])
        dist = _state(s["_distribution"])
        d = _state(dist["_dist"])
        logdet = _find_logdet(dist["_dist"])
        out.append(dict(
            name=str(s["model_name"]),
            data=np.asarray(var["_data"], dtype=np.float64),
            mu=np.asarray(dist["mu"], dtype=np.float64),
            cov=np.asarray(dist["covariance"], dtype=np.float64),
            chol=np.asarray(d["_scale_tri"], dtype=np.float64),
            logdet=None if logdet is None else float(np.asarray(logdet)),
        ))
    return out


def recover_y_mean(data, mu_gold, n_iters=2500, steps=8, tol=2e-10, log=print):
    import scipy.linalg as sla
    X, y, nu = go.prologue(data)
    best = None
    for k in range(steps + 1):
        mu, cov, hyp, st = go.gp_fit(X, y, nu, n_iters, return_state=True)
        err = np.max(np.abs(mu - mu_gold))
        log(f"   step {k}: max|mu-mu_gold|={err:.3e}  hyp={hyp}")
        if best is None or err < best[0]:
            best = (err, y.copy(), mu, cov, hyp)
        if err < tol or k == steps:
            break
        K0, _ = go.kernel_matrix(X, hyp[0], hyp[1])
        N = X.shape[0]
        L = np.linalg.cholesky(K0 + go.JITTER * np.eye(N))
        A = sla.solve_triangular(L, K0, lower=True)
        Sstar = np.linalg.inv(np.eye(N) + (L.T / nu[None, :]) @ L)
        Tm = A.T @ Sstar @ (L.T / nu[None, :])
        y = y + np.linalg.solve(Tm, mu_gold - mu)
    err, y, mu, cov, hyp = best
    log(f"   kept the iterate with max|mu-mu_gold|={err:.3e}")
    return y, mu, cov, hyp


def extract_one(job):
    tag, pkl, idx, steps = job
    from threadpoolctl import threadpool_limits
    lines = []
    log = lambda s: (lines.append(s), print(f"[{tag}] {s}", flush=True))
    with threadpool_limits(1):
        mem = load_members(os.path.join(REF, pkl))[idx]
        R, N = mem["data"].shape
        log(f"{mem['name']}  R={R} N={N} logdet={mem['logdet']}")
        # sanity: the pickled Cholesky/logdet are those of the pickled covariance
        Lc, ld = go.mvn_factor(mem["cov"])
        log(f"   chol check {np.max(np.abs(Lc - mem['chol'])):.2e}  logdet check "
            f"{abs(ld - (mem['logdet'] if mem['logdet'] is not None else ld)):.2e}")
        y, mu, cov, hyp = recover_y_mean(mem["data"], mem["mu"], steps=steps, log=log)
        log(f"   final max|cov-cov_gold|={np.max(np.abs(cov - mem['cov'])):.3e} "
            f"(cov scale {np.max(np.abs(mem['cov'])):.3e})")
        np.savez_compressed(
            os.path.join(OUT, f"gpfit_{tag}.npz"),
            name=mem["name"], data=mem["data"], mu_gold=mem["mu"], cov_gold=mem["cov"],
            chol_gold=mem["chol"], logdet_gold=np.float64(mem["logdet"] if mem["logdet"] is not None else ld),
            y_mean=y, hyp=hyp, n_iters=np.int64(2500),
        )
    return tag


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--jobs", type=int, default=1, help="members extracted side by side (one BLAS thread each)")
    ap.add_argument("--missing", action="store_true", help="only members whose fixture does not exist yet")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    jobs = [(tag, pkl, idx, args.steps) for tag, pkl, idx in SELECTION
            if (not args.only or args.only == tag)
            and not (args.missing and os.path.exists(os.path.join(OUT, f"gpfit_{tag}.npz")))]
    if args.jobs > 1:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(args.jobs) as pool:
            for tag in pool.imap_unordered(extract_one, jobs):
                print(f"[{tag}] written", flush=True)
    else:
        for job in jobs:
            extract_one(job)


if __name__ == "__main__":
    main()
