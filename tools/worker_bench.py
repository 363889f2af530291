import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
names = {12: "shipped mma (two chains, fp64 sign, fp64 merge)", 13: "one chain, integer sign, reads pinned", 14: "two updates interleaved",
         15: "one update, next operands prefetched", 16: "two chains, integer sign, reads pinned"}
for chain in (0, 512):
    for waves in (2, 4, 8):
        print(f"--- {waves} waves{', wave 0 factorising, wave 4 idle' if chain else ''}: cycles per tile update of wave 1")
        for v in (12, 13, 14, 15, 16):
            ctx.bench_diag16(v | chain, waves, 40)
            c = ctx.bench_diag16(v | chain, waves, 400)
            print(f"  {names[v]:52s} {c:8.0f}", flush=True)
