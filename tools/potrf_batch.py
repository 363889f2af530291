import sys; sys.path.insert(0, "/root/repo")
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
for nb in (16, 32, 48, 64):
    print(nb, ctx.bench_potrf(nb, 2048, 10), flush=True)
