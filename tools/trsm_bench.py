"""Times the blocked triangular solves of csrc/driver.cpp (bayens_bench_trsm): the fused 64..256-wide panel launch alone
and the whole recursion, as TFLOP/s on the useful flop (m n^2; a third of it with a lower-triangular right-hand side)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L

ctx = L.Context(0)
nb = 32
for mode, name in [(0, "X L^T = B"), (1, "X L = B")]:
    for m, n in [(2048, 64), (2048, 128), (2048, 256), (1024, 256), (256, 256), (2048, 512), (2048, 1024), (2048, 2048), (64, 2048)]:
        us = ctx.bench_trsm(nb, m, n, mode, False, 20)
        print(f"{name:10s} nb={nb} m={m:5d} n={n:5d}: {us:8.1f} us  {nb * m * n * n / us / 1e6:6.2f} TFLOP/s", flush=True)
us = ctx.bench_trsm(nb, 2048, 2048, 1, True, 20)
print(f"X L = B    nb={nb} lower 2048 (triangular inverse): {us:8.1f} us  {nb * 2048.0 ** 3 / 3 / us / 1e6:6.2f} TFLOP/s")
