"""Generates the NetCDF-4 reader fixtures (run where h5py exists: /opt/conda/bin/python3.9 in the build container).

1. nc_synthetic.nc - a small HDF5 file written with h5py that has the three storage features of the reference's data
   files: a contiguous 2-D float64 variable, an unlimited `time` coordinate in one-element chunks, and a variable
   with more than eight attributes (dense attribute storage: fractal heap + v2 B-tree), plus a deflate+shuffle
   compressed copy of the variable to exercise the filter pipeline.
2. nc_expected.npz - the arrays h5py reads back from that file AND from one of the reference's own data files
   (experiments/data/gmst/ssp126/IPSL_IPSL-CM6A-LR_ssp126_gmst.nc, copied next to this script as a data fixture:
   vlen-string realisation labels, chunked time, contiguous tas)."""
import os
import shutil

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/experiments/data/gmst/ssp126/IPSL_IPSL-CM6A-LR_ssp126_gmst.nc"

rng = np.random.default_rng(0)
tas = 280.0 + rng.standard_normal((3, 24)).cumsum(axis=1)
time = 15.5 + 30.4375 * np.arange(24)
path = os.path.join(HERE, "nc_synthetic.nc")
with h5py.File(path, "w", libver=("earliest", "v110"), track_order=True) as f:
    t = f.create_dataset("time", data=time, maxshape=(None,), chunks=(1,), track_order=True)
    for k, v in [("units", "hours since 1850-01-16 12:00:00"), ("calendar", "proleptic_gregorian"), ("axis", "T"),
                 ("bounds", "time_bnds"), ("standard_name", "time"), ("long_name", "time"), ("a1", "x"), ("a2", "y"),
                 ("a3", "z"), ("a4", "w")]:
        t.attrs.create(k, np.bytes_(v))
    t.make_scale("time")
    r = f.create_dataset("realisation", data=np.arange(3, dtype=np.int64))
    r.make_scale("realisation")
    v = f.create_dataset("tas", data=tas)
    v.attrs.create("_FillValue", np.array([np.nan]))
    v.dims[0].attach_scale(r)
    v.dims[1].attach_scale(t)
with h5py.File(os.path.join(HERE, "nc_synthetic_deflate.nc"), "w") as f:
    f.create_dataset("tas", data=tas, chunks=(2, 16), compression="gzip", shuffle=True, fletcher32=True)

shutil.copyfile(REF, os.path.join(HERE, "nc_reference_IPSL_ssp126_gmst.nc"))
with h5py.File(REF, "r") as h:
    ref_tas, ref_time = h["tas"][...], h["time"][...]
    ref_units, ref_cal = h["time"].attrs["units"].decode(), h["time"].attrs["calendar"].decode()
np.savez(os.path.join(HERE, "nc_expected.npz"), tas=tas, time=time, ref_tas=ref_tas, ref_time=ref_time,
         ref_units=ref_units, ref_cal=ref_cal)
print("ok", ref_tas.shape, ref_units, ref_cal)
