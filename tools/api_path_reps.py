"""Per-repetition wall time of the plug-in API step (ModelCollection.fit -> LogLikelihoodWeight -> Barycentre), 32 x 2000."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesian_ensembling_amd as es
from bayesian_ensembling_amd.labelled import DataArray
from bayesian_ensembling_amd.synthetic import synthetic_ensemble
M, N, R, T = 32, 2000, 8, 2
data, obs = synthetic_ensemble(M, N, R=R, R_obs=3, seed=0xB4E5)
tcoord = np.arange(N)
mk = lambda a: DataArray(a, ("realisation", "time"), {"realisation": np.arange(a.shape[0]), "time": tcoord})
mc = es.ModelCollection([es.ProcessModel(mk(data[m]), f"member{m}") for m in range(M)])
ob = es.ProcessModel(mk(obs), "obs")
for bar in (False, True, False):
    for rep in range(6):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            mc.fit(model=es.GPDTW1D(), n_optim_nits=T, progress_bar=bar)
            t1 = time.perf_counter()
            w = es.LogLikelihoodWeight()(mc, ob, standardisation_constant=1e-3)
            t2 = time.perf_counter()
            post = es.Barycentre()(mc, w)
            t3 = time.perf_counter()
        print(f"bar={bar} rep {rep}: fit {1e3*(t1-t0):7.1f}  weight {1e3*(t2-t1):6.1f}  barycentre {1e3*(t3-t2):6.1f}  total {1e3*(t3-t0):7.1f} ms", flush=True)
