"""Golden vectors for the DTW-barycentre front-end - TEST INFRASTRUCTURE, build container only.

ensembles/dtwa.py is the one reference module that imports stand-alone (NumPy only).  It is loaded here by file
path with importlib (nothing else of the reference is imported) and run on seeded inputs; inputs and outputs are
stored as tests/golden/dba_*.npz.  The reference source is not copied anywhere.

Usage: python oracle/make_dba_golden.py
"""
import importlib.util
import os

import numpy as np

REF = "/root/reference/ensembles/dtwa.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    spec = importlib.util.spec_from_file_location("ref_dtwa", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cases = [("a", 4, 30, 10, 1), ("b", 3, 57, 10, 2), ("c", 8, 64, 5, 3), ("d", 5, 100, 10, 4), ("e", 2, 12, 3, 5)]
    for tag, R, N, iters, seed in cases:
        rng = np.random.default_rng(seed)
        t = np.linspace(0, 1, N)
        series = np.stack([np.sin(2 * np.pi * (t + 0.05 * rng.standard_normal())) * (1 + 0.1 * rng.standard_normal())
                           + 0.15 * rng.standard_normal(N) for _ in range(R)])
        cm = np.zeros((N, N))
        dm = np.zeros((N, N))
        medoid = mod.approximate_medoid_index(list(series), cm, dm)
        d2 = np.array([[mod.squared_DTW(series[i], series[j], cm, dm) for j in range(R)] for i in range(R)])
        center = mod.performDBA(list(series), n_iterations=iters)
        np.savez_compressed(os.path.join(OUT, f"dba_{tag}.npz"), series=series, n_iterations=np.int64(iters),
                            medoid=np.int64(medoid), sq_dtw=d2, center=center)
        print(tag, R, N, iters, "medoid", medoid, "center[:3]", center[:3])


if __name__ == "__main__":
    main()
