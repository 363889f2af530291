import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L
from bayesian_ensembling_amd.synthetic import synthetic_ensemble
ctx = L.Context(0)
data, _ = synthetic_ensemble(32, 2000, R=8)
for rep in range(2):
    t = time.time(); c, m = ctx.dba_batch(data, 10); dt = time.time() - t
    print("dba_batch B=32 R=8 N=2000, 10 iterations: %.3f s wall; medoids %s; max|center - mean| %.3f" % (dt, m[:8], np.max(np.abs(c - data.mean(axis=1)))), flush=True)
