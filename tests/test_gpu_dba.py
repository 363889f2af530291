"""DTW-barycentre front-end: the HIP kernels against the golden vectors of the reference's own
ensembles/dtwa.py (bit-exact: integer/compare work and un-fused fp64 adds in the reference's order) and against the
oracle on seeded inputs; GPDTW1D(target='dba')."""
import os

import numpy as np
import pytest

import bayesian_ensembling_amd as es
from bayesian_ensembling_amd.labelled import DataArray
from conftest import GOLDEN, member_inputs, relerr
from oracle import dba_oracle as do
from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu
FILES = sorted(f for f in os.listdir(GOLDEN) if f.startswith("dba_"))


@pytest.mark.parametrize("fname", FILES)
def test_dba_matches_the_reference_bit_for_bit(ctx, fname):
    g = np.load(os.path.join(GOLDEN, fname))
    center, medoid = ctx.dba_batch(g["series"][None], int(g["n_iterations"]))
    assert medoid[0] == int(g["medoid"])
    assert np.array_equal(center[0], g["center"])
    assert np.array_equal(es.performDBA(list(g["series"]), n_iterations=int(g["n_iterations"])), g["center"])


def test_dba_batch_against_the_oracle(ctx):
    data, _ = go.synthetic_ensemble(5, 150, R=6, seed=77)
    center, medoid = ctx.dba_batch(data, 4)
    for m in range(5):
        assert medoid[m] == do.medoid_index(list(data[m]))
        assert np.array_equal(center[m], do.perform_dba(list(data[m]), 4))


def test_dba_zero_iterations_and_identical_series(ctx):
    x = np.linspace(0, 1, 40) ** 2
    series = np.stack([x, x, x])
    c0, med = ctx.dba_batch(series[None], 0)
    assert med[0] == 0 and np.array_equal(c0[0], x)          # ties: the first candidate wins (strict <)
    c3, _ = ctx.dba_batch(series[None], 3)
    assert np.array_equal(c3[0], do.perform_dba(list(series), 3))


def test_gpdtw1d_with_dba_target(ctx):
    rng = np.random.default_rng(3)
    N = 48
    def da(): return DataArray(np.sin(np.linspace(0, 3, N))[None] + 0.2 * rng.standard_normal((4, N)),
                               ("realisation", "time"), {"realisation": np.arange(4), "time": np.arange(N)})
    mc = es.ModelCollection([es.ProcessModel(da(), "a"), es.ProcessModel(da(), "b")])
    mc.fit(es.GPDTW1D(target="dba"), n_optim_nits=2)
    for pm in mc.models:
        d = pm.model_data.values
        X, _, nu = go.prologue(d)
        mo, co, _ = go.gp_fit(X, do.perform_dba(list(d), 10), nu, 2)
        assert relerr(pm.distribution.mu, mo) < 1e-9 and relerr(pm.distribution.covariance, co) < 1e-9
