"""The NetCDF-4 / HDF5 reader (bayesian_ensembling_amd/nc4.py) against arrays h5py read from the same files
(tests/golden/make_nc_fixtures.py): one of the reference's own data files and two synthetic files covering dense
attribute storage, one-element chunks of an unlimited dimension, and the deflate + shuffle + fletcher32 pipeline."""
import os

import numpy as np
import pytest

import bayesian_ensembling_amd as es
from bayesian_ensembling_amd import nc4
from conftest import GOLDEN

EXP = np.load(os.path.join(GOLDEN, "nc_expected.npz"))


def test_reads_a_reference_data_file_like_xarray_open_dataarray():
    da = nc4.open_dataarray(os.path.join(GOLDEN, "nc_reference_IPSL_ssp126_gmst.nc"))
    assert da.dims == ("realisation", "time") and da.name == "tas"
    assert np.array_equal(da.values, EXP["ref_tas"])
    t = np.asarray(da.coords["time"])
    assert np.array_equal(t, nc4.decode_cf_time(EXP["ref_time"], str(EXP["ref_units"]), str(EXP["ref_cal"])))
    assert str(t[0]).startswith("2015-01-16T12:00") and str(t[-1]).startswith("2100-12-16T12:00")
    assert np.array_equal(da.coords["realisation"], np.arange(6))        # vlen-string labels -> positions


def test_dense_attributes_and_chunked_unlimited_time():
    da = nc4.open_dataarray(os.path.join(GOLDEN, "nc_synthetic.nc"))
    assert da.dims == ("realisation", "time")
    assert np.array_equal(da.values, EXP["tas"])
    # `units` is the first of twelve attributes of `time`: only reachable through the fractal heap
    expect = nc4.decode_cf_time(EXP["time"], "hours since 1850-01-16 12:00:00", "proleptic_gregorian")
    assert np.array_equal(np.asarray(da.coords["time"]), expect)
    assert np.array_equal(da.coords["realisation"], np.arange(3))


def test_filter_pipeline():
    da = nc4.open_dataarray(os.path.join(GOLDEN, "nc_synthetic_deflate.nc"))
    assert np.array_equal(da.values, EXP["tas"])


def test_cf_time_units():
    t = nc4.decode_cf_time([0.0, 1.5], "days since 1850-01-01", "gregorian")
    assert str(t[1]) == "1850-01-02T12:00:00.000000000"
    t = nc4.decode_cf_time([12.0], "hours since 2015-01-16T12:00:00", "proleptic_gregorian")
    assert str(t[0]).startswith("2015-01-17T00:00")
    with pytest.raises(NotImplementedError):
        nc4.decode_cf_time([0.0], "days since 1850-01-01", "360_day")
    with pytest.raises(ValueError):
        nc4.decode_cf_time([0.0], "fortnights after the flood")


def test_rejects_what_it_is_not(tmp_path):
    p = tmp_path / "x.nc"
    p.write_bytes(b"CDF\x01" + b"\0" * 600)          # NetCDF-3 classic: not HDF5
    with pytest.raises(ValueError):
        nc4.open_dataarray(str(p))


def test_file_to_annual_anomaly_process_model():
    """The reference's load path (pre_fitting_cmip6models.py:43-49): file -> ProcessModel -> annual anomaly."""
    da = es.open_dataarray(os.path.join(GOLDEN, "nc_reference_IPSL_ssp126_gmst.nc"))
    pm = es.ProcessModel(da, "IPSL-CM6A-LR")
    clim = np.stack([da.values[:, m::12].mean(axis=1) for m in range(12)]).mean(axis=1)
    anom = pm.calculate_anomaly(climatology=clim, resample_freq="Y")
    assert anom.model_data.shape == (6, 86)           # 2015-2100: the N=86 horizon of BASELINE config 5
    ref = (da.values.reshape(6, 86, 12) - clim[None, None, :]).mean(axis=2)
    assert np.allclose(anom.model_data.values, ref, rtol=0, atol=1e-12)
