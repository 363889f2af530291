"""ensembles/dtwa.py on the GPU: `performDBA(series, n_iterations=10)` - DTW barycentre averaging with a medoid start
(the reference exports it from its package root, ensembles/__init__.py:3).  Equal-length series, at most 50 of them
(beyond 50 the reference draws a random subset for the medoid).  Results are bit-identical to the reference's."""
from __future__ import annotations

import numpy as np

from . import _lib


def performDBA(series, n_iterations=10):
    series = np.asarray([np.asarray(s, dtype=np.float64) for s in series])
    if series.ndim != 2:
        raise NotImplementedError("performDBA: the device path takes series of equal length")
    center, _ = _lib.default_context().dba_batch(series[None], n_iterations)
    return center[0]
