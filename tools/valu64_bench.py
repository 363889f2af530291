"""Throughput of fp64 vector work on one CU (bayens_bench_diag16 kinds 9-11): cycles per wave-instruction with 1, 4 and 8
waves resident.  python tools/valu64_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
for kind, name, per in ((9, "64 v_fma_f64 in 8 independent chains", 64), (10, "8 independent exp_neg (+add)", 8), (11, "8 independent sqrt_pos (+add)", 8)):
    for waves in (1, 4, 8):
        ctx.bench_diag16(kind, waves, 20)
        c = ctx.bench_diag16(kind, waves, 400)
        print(f"{name:40s} {waves} waves: {c:8.1f} cycles per repetition = {c / per:6.1f} per item per wave", flush=True)
