"""Run one batched Cholesky (32 x 2048) twice; under rocprofv3 --kernel-trace the second run's dispatch sequence
(tools/rocpd_seq.py) shows where the factorisation's time goes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
nb, n = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(0)
M = rng.standard_normal((n, n))
S1 = M @ M.T + n * np.eye(n)
S = np.broadcast_to(S1, (nb, n, n)).copy()
for rep in range(2):
    Lg, info = ctx.test_potrf(S)
print("info", info.max(), "err", np.abs(Lg[0] - np.linalg.cholesky(S1)).max())
