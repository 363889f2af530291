"""NN / NT / TN variants of gemm_f64_mfma on the fit's solve shapes (m rows x n columns, k deep), 32 members."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
TA, TB, TRI, KTRIM = L.GEMM_TA, L.GEMM_TB, L.GEMM_TRI_OUT, L.GEMM_KTRIM_TN_LOWER
for m, n, k in [(2048, 2048, 2048), (2048, 1024, 1024), (2048, 512, 512), (2048, 256, 256), (1024, 1024, 1024), (1024, 512, 512)]:
    row = []
    for name, fl in [("NN", 0), ("NT", TB), ("TN", TA)]:
        ms = ctx.bench_gemm(32, m, n, k, fl, reps=5)
        row.append(f"{name} {ms*1e3:8.1f} us {2.0*m*n*k*32/(ms*1e-3)/1e12:6.2f} TF/s")
    print(f"{m}x{n}x{k}: " + " | ".join(row), flush=True)
