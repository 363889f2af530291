"""Repeats the three routes' fits and the step of the headline configuration and compares the results of the repetitions BIT FOR BIT
(a data race or an uninitialised read shows up as a difference between two runs of the same launch).  GPU only; prints one line per case."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from bayesian_ensembling_amd import _lib as L      # noqa: E402


def synthetic(M, N, R, seed):
    rng = np.random.default_rng(seed)
    tau = np.linspace(0.0, 1.0, N)
    data = 1.5 * (0.8 + 0.4 * rng.random((M, 1, 1))) * tau ** 2 + 0.036 * rng.standard_normal((M, R, N)).cumsum(axis=2)
    return np.ascontiguousarray(np.swapaxes(data, 1, 2)), data.mean(axis=1), data.var(axis=1)


def main():
    ctx = L.Context(0)
    cases = [("small  (fused, N = 86)", 512, 86, 8, 2, 6), ("medium (fused, N = 165)", 20, 165, 9, 2500, 4),
             ("medium (fused, N = 165, 140 members)", 140, 165, 9, 200, 4), ("general (N = 700)", 8, 700, 6, 6, 6),
             ("general (N = 2000, 32 members)", 32, 2000, 8, 2, 4)]
    ok = True
    for name, M, N, R, T, reps in cases:
        X, y, nu = synthetic(M, N, R, seed=N + M)
        first, same = None, True
        for _ in range(reps):
            out = ctx.gp_fit_batch(X, y, nu, T, want_loss=False)
            blob = (out["mu"].tobytes(), out["cov"].tobytes(), out["chol"].tobytes(), out["hyp"].tobytes(), out["logdet"].tobytes())
            if first is None:
                first = blob
            same = same and blob == first
            del out
        ok = ok and same
        print(f"{name:40s} M={M:4d} N={N:5d} T={T:5d}: {reps} repetitions {'bit-identical' if same else 'DIFFER'}", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
