"""Container behaviour mirrored from the reference's tests/test_data.py:43-99 (shapes of statistics,
anomaly round trip, realisation iterator, collection counts/iteration) on the xarray-free DataArray."""
import numpy as np
import pytest

import bayesian_ensembling_amd as es
from bayesian_ensembling_amd.labelled import DataArray, concat


def make_da(n_dims, n_reals=3, periods=480, seed=None):
    rng = np.random.default_rng(seed)
    time = np.arange("1960-01", periods and f"{1960 + periods // 12}-{periods % 12 + 1:02d}", dtype="datetime64[M]")
    time = (time + 1).astype("datetime64[D]") - np.timedelta64(1, "D")     # month ends, like freq='M'
    dims = ["realisation", "time", "lon", "lat"][: n_dims + 1]
    shape = [n_reals, len(time), 4, 5][: n_dims + 1]
    coords = {"realisation": np.arange(n_reals), "time": time, "lon": np.arange(4), "lat": np.arange(5)}
    return DataArray(rng.random(shape), dims, {k: coords[k] for k in dims})


@pytest.mark.parametrize("n_dims", [1, 2, 3])
def test_process_model(n_dims):
    model = es.ProcessModel(make_da(n_dims), "model_name")
    assert model.model_data.ndim == n_dims + 1
    assert model.model_mean.size == 1 and model.model_std.size == 1
    assert model.max_val.size == 1 and model.min_val.size == 1
    anomaly = model.calculate_anomaly()
    again = model.calculate_anomaly(climatology=anomaly.climatology)
    assert again.model_data.shape == model.model_data.shape
    assert np.allclose(again.model_data.values, anomaly.model_data.values)
    n = 0
    for realisation in model:
        assert isinstance(realisation, DataArray)
        n += 1
    assert n == model.n_realisations
    annual = model.calculate_anomaly(resample_freq="Y")
    assert annual.model_data.shape[1] == 40


@pytest.mark.parametrize("n_dims", [1, 2, 3])
@pytest.mark.parametrize("n_models", [1, 2, 3])
def test_model_collection(n_dims, n_models):
    mc = es.ModelCollection([es.ProcessModel(make_da(n_dims), f"m{i}") for i in range(n_models)])
    assert np.size(mc.max_val) == 1 and np.size(mc.min_val) == 1
    assert mc.number_of_models == n_models and len(mc.model_names) == n_models
    assert sum(1 for m in mc if isinstance(m, es.ProcessModel)) == n_models
    assert len(mc) == n_models and mc[0].model_name == "m0"


def test_time_axis_mismatch_is_fixed_with_a_warning():
    a, b = make_da(1, periods=24), make_da(1, periods=24)
    b.coords["time"] = b.coords["time"] + np.timedelta64(1, "D")
    with pytest.warns(UserWarning):
        mc = es.ModelCollection([es.ProcessModel(a, "a"), es.ProcessModel(b, "b")])
    assert np.all(mc[0].time.values == mc[1].time.values)


def test_process_model_requires_realisation_first():
    with pytest.raises(AssertionError):
        es.ProcessModel(DataArray(np.zeros((3, 4)), ("time", "realisation")), "bad")


def test_dataarray_named_broadcast_and_reductions():
    a = DataArray(np.arange(6.0).reshape(2, 3), ("model", "time"), {"model": np.array(["x", "y"]), "time": np.arange(3)})
    s = a.sum("model")
    w = a / s
    assert w.dims == ("model", "time") and np.allclose(w.sum("model").values, 1.0)
    assert np.allclose(a.sel(model="y").values, [3, 4, 5]) and a.mean("time").shape == (2,)
    rows = [a.isel(model=i).assign_coords(model=n) for i, n in enumerate(["p", "q"])]
    c = concat(rows, dim="model")
    assert c.dims == ("model", "time") and list(c.coords["model"]) == ["p", "q"]
    e = s.expand_dims(time2=np.arange(4), axis=0)
    assert e.shape == (4, 3)


def test_distribution_fields_are_assignable_like_the_references_dataclass():
    """ensembles/data.py:18-24 is a plain @dataclass: `dist.mu = ...` works, and it has a generated __repr__ / __eq__."""
    from bayesian_ensembling_amd import distributions as dx
    from bayesian_ensembling_amd.data import Distribution
    blank = DataArray(np.full(4, np.nan), ("time",), {"time": np.arange(4)})
    d = Distribution(mu=np.arange(4.0), covariance=np.full(4, 2.0), dim_array=blank, dist_type=dx.MultivariateNormalDiag)
    e = Distribution(mu=np.arange(4.0), covariance=np.full(4, 2.0), dim_array=blank, dist_type=dx.MultivariateNormalDiag)
    assert d == e and "MultivariateNormalDiag" in repr(d)
    d.mu = np.ones(4)
    assert np.array_equal(d.mean.values, np.ones(4)) and d != e
    d.covariance = np.full(4, 3.0)
    assert np.array_equal(d._dist.variance(), np.full(4, 9.0))            # scale quirk re-derived on assignment


def test_ragged_layout_and_field_launch_size_are_pure_host_logic():
    """The layout `bayens_gp_fit_batch_ragged{,_f32}` takes (members' (N, R_b) blocks one behind the other, offsets in elements) and
    the number of general-path fits a per-gridpoint field puts into one launch (workspace budget / padded matrices per fit)."""
    import numpy as np
    from bayesian_ensembling_amd import _lib
    from bayesian_ensembling_amd.models import GPDTW1D
    Xs = [np.arange(12.0).reshape(4, 3), np.arange(20.0).reshape(4, 5) + 100, np.arange(4.0).reshape(4, 1) - 7]
    R, off, X = _lib.Context.ragged_layout(Xs)
    assert R.tolist() == [3, 5, 1] and off.tolist() == [0, 12, 32] and X.shape == (36,)
    for r, o, x in zip(R, off, Xs):
        assert np.array_equal(X[int(o):int(o) + 4 * int(r)].reshape(4, int(r)), x)
    with pytest.raises(ValueError):
        _lib.Context.ragged_layout([np.zeros((4, 2)), np.zeros((5, 2))])
    m = GPDTW1D(per_gridpoint=True)
    for N, itemsize in ((1980, 8), (1980, 4), (300, 8)):
        Np = -(-N // 64) * 64
        per_launch = int(m.FIELD_WS_BYTES // (m.FIELD_WS_MATRICES * Np * Np * itemsize))
        assert per_launch >= 1 and per_launch * m.FIELD_WS_MATRICES * Np * Np * itemsize <= m.FIELD_WS_BYTES
