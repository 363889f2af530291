"""Micro-benchmark of the 16x16 diagonal factorisation of the fused small-member kernel and of the dependent chains it
is made of (bayens_bench_diag16).  python tools/diag16_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
names = {0: "diag16 (factor + inverse, tile restored)", 1: "64 dependent v_fma_f64", 2: "16 dependent v_rsq_f64 (+fma)",
         3: "16 dependent refined rsqrt (+add)", 4: "16 dependent MFMA f64 16x16x4", 5: "16 x (readlane of MFMA result -> next MFMA)",
         6: "tile LDS load + store + read"}
for waves in (1, 8):
    for busy in (0, 256):
        if busy and waves == 1:
            continue
        print(f"--- {waves} wave(s) in the workgroup{', partners issuing MFMAs' if busy else ''}")
        for v in range(7):
            ctx.bench_diag16(v | busy, waves, 20)
            c = ctx.bench_diag16(v | busy, waves, 200)
            per = {1: 64, 2: 16, 3: 16, 4: 16, 5: 16}.get(v)
            print(f"  {names[v]:48s} {c:9.0f} cycles" + (f"  = {c / per:6.1f} per link" if per else ""), flush=True)

print("--- product loops, cycles per 6-step tile product per wave (all waves of the workgroup busy)")
for waves in (1, 4, 8):
    for v, nme in ((7, "acc_run from LDS"), (8, "24 MFMAs from registers")):
        ctx.bench_diag16(v, waves, 20)
        c = ctx.bench_diag16(v, waves, 200)
        print(f"  {waves} waves  {nme:28s} {c:8.0f} cycles per product-of-6-steps = {c / 24:6.1f} per MFMA", flush=True)
