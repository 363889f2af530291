import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
us, st = ctx.bench_leaf(32, 50)
print("copy+leaf per rep: %.2f us" % us)
names = ["load"] + [f"bj{b}:{p}" for b in range(4) for p in ("P1", "P2", "P3")] + ["P4", "store"]
d = [int(st[i + 1]) - int(st[i]) for i in range(14)]
print("cycles between stamps (s_memtime ticks = 100MHz? or shader clock):")
for n_, v in zip(names[1:], d):
    print(f"  {n_:8s} {v}")
print("total", int(st[14]) - int(st[0]))
