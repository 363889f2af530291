"""Fixed-overhead fit of the GEMM: time(k) = a + b k at fixed m, n (beta from BAYENS_BENCH_BETA)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
for (m, n) in [(2048, 256), (2048, 2048)]:
    for fl, name in [(L.GEMM_TB, "NT"), (0, "NN")]:
        ts = []
        for k in (128, 256, 512, 1024, 2048):
            ms = ctx.bench_gemm(32, m, n, k, fl, reps=10)
            ts.append((k, ms))
        b = (ts[-1][1] - ts[1][1]) / (ts[-1][0] - ts[1][0])
        a = ts[1][1] - b * ts[1][0]
        print(f"beta={os.environ.get('BAYENS_BENCH_BETA','0.5')} {name} m={m} n={n}: " + "  ".join(f"k={k}:{ms*1e3:.0f}us" for k, ms in ts) +
              f"  -> fixed {a*1e3:.0f} us, {b*1e3*16:.2f} us per 16-k stage, asymptotic {2.0*m*n*32/(b*1e-3)/1e12:.1f} TFLOP/s", flush=True)
