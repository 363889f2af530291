"""Time the batched blocked Cholesky alone (device-generated SPD input), for several orders."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
for n in (2048, 1024, 512, 256, 128):
    us = ctx.bench_potrf(32, n, 10)
    print(f"potrf 32 x {n}: {us:9.1f} us  {32 * n**3 / 3 / us / 1e6:6.2f} TFLOP/s", flush=True)
