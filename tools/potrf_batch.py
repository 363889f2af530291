"""One blocked Cholesky of nb 2048 x 2048 members against nb (bayens_bench_potrf): the time grows far slower than the batch - the
factorisation is bound by its chain of small kernels - which is why the first iteration factors its two matrices as one batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L

ctx = L.Context(0)
for nb in (16, 32, 48, 64):
    us = ctx.bench_potrf(nb, 2048, 10)
    print(f"potrf nb={nb:3d} n=2048: {us:8.1f} us  {us / nb:6.1f} us per member  {nb * 2048.0 ** 3 / 3 / us / 1e6:6.2f} TFLOP/s", flush=True)
