"""Timeline of the fused 256 x 256 diagonal-block Cholesky (csrc/potrf256.hip): per-wave cycle stamps of workgroup 0."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
for nb in (1, 32, 64):
    us = ctx.bench_potrf(nb, 256, 20)
    print(f"potrf {nb} x 256: {us:7.1f} us", flush=True)
st = ctx.potrf256_stamps().astype(np.int64)
names = ["start", "in", "sync", "diag0", "col0", "col7", "cols", "out", "sync", "inv"]
print("wave  " + " ".join(f"{n:>7s}" for n in names))
for w in range(8):
    print(f"  {w}   " + " ".join(f"{(st[w, k] - st[w, 0]) if st[w, k] else 0:7d}" for k in range(10)))
