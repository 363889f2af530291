"""Per-gridpoint fit of a field at the reference's monthly horizon (N = 1980 time steps, experiments/data/**/*.nc) through the plug-in
classes: GPDTW1D(per_gridpoint=True) -> LogLikelihoodWeight -> Barycentre.  The fits run on the general path with the reduced export
(the covariance stays in the workspace).  Prints one JSON line.  GPU only."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bayesian_ensembling_amd as es                      # noqa: E402
from bayesian_ensembling_amd.labelled import DataArray    # noqa: E402


def main(M=4, R=5, N=1980, n_lat=4, n_lon=8, nits=2, dtype="float64"):
    rng = np.random.default_rng(0)
    tau = np.linspace(0.0, 1.0, N)[None, :, None, None]
    coords = {"time": np.arange(N), "latitude": np.linspace(-60, 60, n_lat), "longitude": np.linspace(0, 315, n_lon)}
    dims = ("realisation", "time", "latitude", "longitude")

    def da(amp, n_real):
        vals = 1.5 * amp * tau ** 2 + 0.036 * rng.standard_normal((n_real, N, n_lat, n_lon)).cumsum(axis=1)
        return DataArray(vals, dims, dict(coords, realisation=np.arange(n_real)))
    mc = es.ModelCollection([es.ProcessModel(da(0.8 + 0.1 * m, R), f"model{m}") for m in range(M)])
    obs = es.ProcessModel(da(1.0, 3), "obs")
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        mc.fit(model=es.GPDTW1D(per_gridpoint=True, dtype=dtype), n_optim_nits=nits, progress_bar=False)
        w = es.LogLikelihoodWeight()(mc, obs, standardisation_constant=1e-3)
        post = es.Barycentre()(mc, w)
        np.asarray(post.mu)
        times.append(time.perf_counter() - t0)
    fits = M * n_lat * n_lon
    print(json.dumps({"workload": f"{M} members x {n_lat}x{n_lon} grid points x N={N} (R={R}, n_optim_nits={nits}, {dtype})",
                      "fits": fits, "seconds": round(min(times), 4), "fits_per_s": round(fits / min(times), 1),
                      "finite": bool(np.isfinite(np.asarray(post.mu)).all())}))


if __name__ == "__main__":
    main(dtype=sys.argv[1] if len(sys.argv) > 1 else "float64")
