"""GEMM shape sweep on the GPU box: ms and TFLOP/s (algorithmic) per shape.  python tools/gemm_bench.py [--quick]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L  # noqa: E402

ctx = L.Context(0)
TA, TB, TRI, KTRIM = L.GEMM_TA, L.GEMM_TB, L.GEMM_TRI_OUT, L.GEMM_KTRIM_TN_LOWER
shapes = [
    ("NT full 2048^3 x32", 32, 2048, 2048, 2048, TB),
    ("NT syrk lower 2048^3 x32", 32, 2048, 2048, 2048, TB | TRI),
    ("NN full 2048^3 x32", 32, 2048, 2048, 2048, 0),
    ("TN lower ktrim 2048^3 x32", 32, 2048, 2048, 2048, TA | TRI | KTRIM),
    ("NT 2048x1024x1024 x32", 32, 2048, 1024, 1024, TB),
    ("NT 2048x512x512 x32", 32, 2048, 512, 512, TB),
    ("NT 2048x256x256 x32", 32, 2048, 256, 256, TB),
    ("NT 2048x128x128 x32", 32, 2048, 128, 128, TB),
    ("NT 2048x64x64 x32 (leaf)", 32, 2048, 64, 64, TB),
    ("NN 2048x64x64 x32 (leaf)", 32, 2048, 64, 64, 0),
    ("NT 1024x64x64 x32", 32, 1024, 64, 64, TB),
    ("NT 128x128x64 x32", 32, 128, 128, 64, TB),
    ("NT 64x64x64 x32", 32, 64, 64, 64, TB),
]
if "--big" in sys.argv:
    shapes = shapes[:4]
for name, nb, m, n, k, fl in shapes:
    ms = ctx.bench_gemm(nb, m, n, k, fl, reps=5 if m * n * k > 1e9 else 50)
    f = 2.0 * m * n * k * nb
    if (fl & TRI) and (fl & KTRIM):
        f /= 6
    elif fl & TRI:
        f /= 2
    print(f"{name:32s} {ms * 1e3:10.1f} us  {f / (ms * 1e-3) / 1e12:7.2f} TFLOP/s", flush=True)
