"""cProfile of config 5 through the plug-in classes (fit -> LogLikelihoodWeight -> Barycentre), after a warm-up pass."""
import cProfile, io, os, pstats, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesian_ensembling_amd as es
from bayesian_ensembling_amd import _lib
from bayesian_ensembling_amd.labelled import DataArray
ctx = _lib.default_context()
M, n_lat, n_lon, N, R, T = 32, 36, 48, 86, 8, 2
rng = np.random.default_rng(0xC5)
tau = np.linspace(0.0, 1.0, N)[None, :, None, None]
loc = 1.0 + 0.2 * rng.standard_normal((1, 1, n_lat, n_lon))
coords = {"time": np.arange(N), "latitude": np.linspace(-89, 89, n_lat), "longitude": np.linspace(0, 357.5, n_lon)}
dims = ("realisation", "time", "latitude", "longitude")
def pm(amp, n_real, name):
    vals = 1.5 * amp * loc * tau ** 2 + 0.036 * rng.standard_normal((n_real, N, n_lat, n_lon)).cumsum(axis=1)
    return es.ProcessModel(DataArray(vals, dims, dict(coords, realisation=np.arange(n_real))), name)
mc = es.ModelCollection([pm(0.8 + 0.4 * rng.random(), R, f"member{m}") for m in range(M)])
ob = pm(1.0, 3, "obs")
def step():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        mc.fit(model=es.GPDTW1D(per_gridpoint=True), n_optim_nits=T, progress_bar=False)
        ctx.sync(); t1 = time.perf_counter()
        w = es.LogLikelihoodWeight()(mc, ob, standardisation_constant=1e-3)
        t2 = time.perf_counter()
        post = es.Barycentre()(mc, w)
        _ = np.asarray(post.mu); t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2
for _ in range(2):
    print("ms fit / weights / barycentre:", [round(x * 1e3, 2) for x in step()], flush=True)
pr = cProfile.Profile(); pr.enable(); parts = step(); pr.disable()
print("profiled ms:", [round(x * 1e3, 2) for x in parts])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
