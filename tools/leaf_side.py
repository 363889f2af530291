"""The 64x64 leaf factorisation alone, beside a trailing-update-sized product and beside a copy of the same size
(BAYENS_LEAF_SIDE = 0 / 1 / 2): which of its phases stretch when the side stream is busy."""
import os, sys, subprocess
if len(sys.argv) == 1:
    for k in ("0", "1", "2"):
        env = dict(os.environ, BAYENS_LEAF_SIDE=k)
        print(f"--- BAYENS_LEAF_SIDE={k}", flush=True)
        subprocess.run([sys.executable, __file__, "run"], env=env, check=False)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
ctx = L.Context(0)
us, st = ctx.bench_leaf(32, 20)
names = ["load"] + [f"bj{b}:{p}" for b in range(4) for p in ("P1", "P2", "P3")] + ["P4", "store"]
d = [int(st[i + 1]) - int(st[i]) for i in range(14)]
print("stamped launch, ticks between stamps:", " ".join(f"{n}={v}" for n, v in zip(names[1:], d)), "| total", int(st[14]) - int(st[0]))
print(f"unloaded copy+leaf per rep afterwards: {us:.1f} us")
