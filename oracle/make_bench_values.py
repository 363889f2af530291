"""TEST INFRASTRUCTURE (oracle side): known answers for the long runs bench.py times, so that a bench line can say "and the
result is right" without running the oracle on the GPU box for minutes (tests/golden/bench_values.json).

  n165_T2500   member 0 of bench.mid_size_line's collection (synthetic_ensemble(20, 165, R = 9, seed 0x165)), n_optim_nits = 2500
               - the paper's horizon and iteration count (pre_fitting_cmip6models.py:76-77)
  c2_T500      member 0 of the headline batch (synthetic_ensemble(32, 2000, R = 8, seed 0xB4E5)) at the API default
               n_optim_nits = 500 (ensembles/models.py:167)

per entry: the final (variance, lengthscale), mean(mu), trace(cov), log det chol(cov), all from oracle/gp_oracle.py (fp64 NumPy).
    python oracle/make_bench_values.py [n165_T2500] [c2_T500]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gp_oracle as go  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "bench_values.json")
CASES = {"n165_T2500": dict(M=20, N=165, R=9, seed=0x165, T=2500), "c2_T500": dict(M=32, N=2000, R=8, seed=0xB4E5, T=500)}


def main():
    want = [a for a in sys.argv[1:] if a in CASES] or list(CASES)
    vals = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in want:
        c = CASES[name]
        data, _ = go.synthetic_ensemble(c["M"], c["N"], R=c["R"], seed=c["seed"])
        X, y, nu = go.prologue(data[0])
        t0 = time.time()
        mu, cov, hyp = go.gp_fit(X, y, nu, c["T"])[:3]
        _, ld = go.mvn_factor(cov)
        vals[name] = dict(c, member=0, variance=float(hyp[0]), lengthscale=float(hyp[1]), mean_mu=float(np.mean(mu)),
                          trace_cov=float(np.trace(cov)), logdet=float(ld), oracle_seconds=round(time.time() - t0, 1))
        print(name, vals[name], flush=True)
        json.dump(vals, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
