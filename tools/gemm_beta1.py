"""C += A B^T (beta = 1) on the short-K shapes of the Cholesky and the solves, with and without BAYENS_NO_CACC."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
TB, TRI = L.GEMM_TB, L.GEMM_TRI_OUT
for m, n, k, fl in [(2048, 256, 256, TB), (1792, 1792, 256, TB | TRI), (1536, 1536, 256, TB | TRI), (1024, 1024, 256, TB | TRI),
                    (1792, 256, 256, TB), (2048, 512, 512, TB), (2048, 1024, 1024, TB), (2048, 256, 256, 0), (2048, 1024, 1024, 0)]:
    ms = ctx.bench_gemm(32, m, n, k, fl, reps=10)
    f = 2.0 * m * n * k * 32 * (0.5 * (1 + 64.0 / m) if fl & TRI else 1.0)
    print(f"{m}x{n}x{k} flags {fl}: {ms*1e3:8.1f} us {f/(ms*1e-3)/1e12:6.2f} TF/s", flush=True)
