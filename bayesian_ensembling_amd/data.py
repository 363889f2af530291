"""Containers of the plug-in API: Distribution, ProcessModel, ModelCollection.

Same names, fields, properties and error behaviour as the reference (ensembles/data.py:18-143, 146-352,
355-562) minus plotting; `ModelCollection.fit` additionally batches all members into one device call when
the emulator offers `fit_collection` (members are independent: data.py:391 loops over them serially).
"""
from __future__ import annotations

import pickle
import typing as tp
import warnings
from dataclasses import dataclass

import numpy as np

from . import distributions as dx
from .labelled import DataArray


class Distribution:
    """ensembles/data.py:18-143.  `dist_type(mu, covariance)` is called positionally (data.py:39).

    Same constructor and attributes as the reference's dataclass.  A posterior that came out of a GPU fit is *resident*:
    `_resident = (ResidentBatch, index)` and `mu` / `covariance` are downloaded from HBM the first time they are read."""

    def __init__(self, mu=None, covariance=None, dim_array=None, dist_type=None, _factor=None, _resident=None):
        self._mu, self._covariance = mu, covariance
        self.dim_array, self.dist_type, self._factor, self._resident = dim_array, dist_type, _factor, _resident
        self.__post_init__()

    def __post_init__(self):
        if self._resident is not None and self.dist_type is dx.GridMultivariateNormalFullCovariance:
            self._dist = self.dist_type(*self._resident)             # (ResidentField, member index)
        elif self._resident is not None:
            assert self.dist_type is dx.MultivariateNormalFullCovariance
            self._dist = self.dist_type._from_resident(*self._resident)
        elif self.dist_type is dx.MultivariateNormalFullCovariance and self._factor is not None:
            self._dist = self.dist_type(self._mu, self._covariance, _chol=self._factor[0], _logdet=self._factor[1],
                                        _basis=self._factor[2])
        else:
            self._dist = self.dist_type(self._mu, self._covariance)

    @property
    def mu(self):
        if self._mu is None:
            self._mu = self._dist.mean()
        return self._mu

    @property
    def covariance(self):
        if self._covariance is None:
            self._covariance = self._dist.covariance()
        return self._covariance

    # the reference's Distribution is a plain dataclass (ensembles/data.py:18-24): fields are assignable and it has a
    # generated __repr__ / __eq__; assigning a field here re-derives the distribution like a fresh construction would
    @mu.setter
    def mu(self, value):
        self._materialise()
        self._mu = value
        self.__post_init__()

    @covariance.setter
    def covariance(self, value):
        self._materialise()
        self._covariance = value
        self.__post_init__()

    def _materialise(self):
        """Resident / pre-factored posterior -> plain host fields (before one of them is replaced)."""
        if self._resident is not None and self.dist_type is dx.MultivariateNormalFullCovariance:
            self._mu, self._covariance = np.array(self.mu), np.array(self.covariance)
        elif self._resident is not None:
            raise AttributeError("a per-gridpoint field posterior is read-only")
        self._resident, self._factor = None, None

    def __repr__(self):
        def short(a):
            return None if a is None else f"array(shape={np.shape(a)})"
        where = "resident in HBM" if self._resident is not None else "host"
        return (f"Distribution(mu={short(self._mu) if self._resident is None else '<HBM>'}, "
                f"covariance={short(self._covariance) if self._resident is None else '<HBM>'}, "
                f"dim_array={getattr(self.dim_array, 'shape', None)}, dist_type={getattr(self.dist_type, '__name__', None)}) [{where}]")

    def __eq__(self, other):
        """Equality of two posteriors.  Two views of the SAME resident posterior (a deep copy shares the HBM buffers) are equal
        without touching the device; per-gridpoint fields - which keep marginals only, `covariance` raises - compare by mean
        and marginal variance; everything else by mean and covariance like the reference's dataclass.  (A field pickles as the
        marginal pair MultivariateNormalDiag(mean, stddev) - see __getstate__ - so a save / load round trip changes its
        dist_type and the reloaded object does NOT compare equal to the resident original: what survives is what the
        reference's consumers read, mean() and variance().)"""
        if not isinstance(other, Distribution):
            return NotImplemented
        if self.dist_type is not other.dist_type:
            return False
        if self._resident is not None and other._resident is not None and self._resident[0] is other._resident[0] \
                and self._resident[1] == other._resident[1]:
            return True
        if self.dist_type is dx.GridMultivariateNormalFullCovariance:
            return (np.array_equal(np.asarray(self._dist.mean()), np.asarray(other._dist.mean()))
                    and np.array_equal(np.asarray(self._dist.variance()), np.asarray(other._dist.variance())))
        return (np.array_equal(np.asarray(self.mu), np.asarray(other.mu))
                and np.array_equal(np.asarray(self.covariance), np.asarray(other.covariance)))

    __hash__ = None

    def __getstate__(self):
        """Pickling (ModelCollection.save, data.py:397-399) takes the posterior to the host."""
        state = dict(self.__dict__)
        if self._resident is not None and self.dist_type is dx.GridMultivariateNormalFullCovariance:
            # a field keeps marginals only: it pickles as the dx.Normal-style pair the reference's consumers read
            state.update(_mu=np.array(self._dist.mean()), _covariance=np.array(self._dist.stddev()), _resident=None,
                         dist_type=dx.MultivariateNormalDiag)
            del state["_dist"]
        elif self._resident is not None:
            d = self._dist
            state.update(_mu=np.array(self.mu), _covariance=np.array(self.covariance), _resident=None,
                         _factor=(np.array(d._scale_tri), float(d._logdet), np.array(d._basis)))
            del state["_dist"]
        return state

    def __deepcopy__(self, memo):
        """A resident posterior is immutable device data: a deep copy (PerfectModelTest.run copies the whole collection
        once per held-out member, ensembles/utils.py:196-199) shares the HBM buffers instead of forcing the 64 MB per
        member download that pickling needs; host-side fields are copied as usual."""
        import copy
        if self._resident is None:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                setattr(new, k, copy.deepcopy(v, memo))
            return new
        new = Distribution(dim_array=copy.deepcopy(self.dim_array, memo), dist_type=self.dist_type, _resident=self._resident)
        for k in ("hyperparameters", "losses"):
            if hasattr(self, k):
                setattr(new, k, copy.deepcopy(getattr(self, k), memo))
        memo[id(self)] = new
        return new

    def __setstate__(self, state):
        rebuild = "_dist" not in state
        self.__dict__.update(state)
        if rebuild:
            self.__post_init__()

    def reshape(self, vals, name=False):
        """data.py:41-55"""
        reshaped_vals = np.asarray(vals).reshape(self.dim_array.shape)
        reshaped_array = self.dim_array.copy(data=reshaped_vals)
        if name:
            reshaped_array = reshaped_array.rename(name)
        return reshaped_array

    @property
    def mean(self):
        return self.reshape(self._dist.mean(), name="Distribution mean")

    @property
    def variance(self):
        return self.reshape(self._dist.variance(), name="Distribution variance")

    def sample(self):
        samples = np.asarray(self._dist.sample(seed=np.random.randint(0, 110000)))
        return self.reshape(samples, name="Distribution sample")


@dataclass
class ProcessModel:
    """ensembles/data.py:146-352: the simulation output of one process model, dims (realisation, time, ...)."""
    model_data: DataArray
    model_name: str
    idx: int = 0
    _distribution = None

    def __post_init__(self):
        self.model_data = DataArray.from_any(self.model_data)
        self.model_mean = self.model_data.mean()
        self.model_std = self.model_data.std()
        self.climatology = None
        assert isinstance(self.model_data, DataArray), "Input must be a DataArray"
        assert self.model_data.dims[0] == "realisation"
        assert np.any(~np.isnan(self.model_data.values)), "Input data must not contain NaN"

    @property
    def max_val(self):
        return self.model_data.max()

    @property
    def min_val(self):
        return self.model_data.min()

    @property
    def n_observations(self) -> int:
        raise NotImplementedError

    @property
    def n_realisations(self) -> int:
        return self.model_data.shape[0]

    @property
    def time(self):
        return self.model_data.time

    def calculate_anomaly(self, climatology_dates=("1961-01-01", "1990-12-31"), climatology=False,
                          resample_freq=None):
        """data.py:225-261: anomaly w.r.t. a monthly climatology; optional annual ('Y') resampling."""
        da = self.model_data.copy(deep=True)
        t = np.asarray(da.coords["time"]).astype("datetime64[D]")
        month = t.astype("datetime64[M]").astype(int) % 12
        tax = da.dims.index("time")
        if np.any(climatology) == False:  # noqa: E712  (same test as the reference)
            t0, t1 = np.datetime64(climatology_dates[0]), np.datetime64(climatology_dates[1])
            sel = (t >= t0) & (t <= t1)
            vals = np.moveaxis(da.values, tax, 0)
            clim = np.stack([vals[sel & (month == m)].mean(axis=0) for m in range(12)])   # (12, real, ...)
            clim = clim.mean(axis=1)                                                      # mean over realisation
        else:
            clim = np.asarray(getattr(climatology, "values", climatology))
            assert clim.shape[0] == 12, "Climatology is the incorrect length (must be 12)"
        vals = np.moveaxis(da.values, tax, 0).astype(np.float64)      # (time, real, ...)
        anom = vals - clim[month][:, None, ...]
        tcoord = da.coords["time"]
        if resample_freq:
            if resample_freq not in ("Y", "A", "YE"):
                raise NotImplementedError("only annual resampling is implemented")
            years = t.astype("datetime64[Y]")
            uy = np.unique(years)
            anom = np.stack([anom[years == y].mean(axis=0) for y in uy])
            tcoord = (uy + 1).astype("datetime64[D]") - np.timedelta64(1, "D")   # year-end stamps like xarray
        anom = np.moveaxis(anom, 0, tax)
        coords = dict(da.coords)
        coords["time"] = tcoord
        out = DataArray(anom, da.dims, coords, da.name)
        anomaly_model = ProcessModel(out, self.model_name + " anomaly")
        anomaly_model.climatology = DataArray(clim, ("month",) + tuple(d for d in da.dims if d not in ("time", "realisation")),
                                              {"month": np.arange(1, 13)})
        return anomaly_model

    @property
    def mean_across_realisations(self):
        return self.model_data.mean("realisation")

    @property
    def std_across_realisations(self):
        return self.model_data.std("realisation")

    @property
    def ndim(self):
        return self.model_data.ndim

    @property
    def distribution(self) -> Distribution:
        return self._distribution

    @distribution.setter
    def distribution(self, dist: Distribution):
        self._distribution = dist

    def __len__(self) -> int:
        return self.n_realisations

    def __iter__(self):
        return self

    def __next__(self):
        try:
            if self.idx >= self.n_realisations:
                raise IndexError
            out = self.model_data.isel(realisation=self.idx)
            self.idx += 1
        except IndexError:
            self.idx = 0
            raise StopIteration
        return out


@dataclass
class ModelCollection:
    """ensembles/data.py:355-562."""
    models: tp.List[ProcessModel]
    idx: int = 0

    def __post_init__(self):
        self.check_time_axes()

    def __iter__(self):
        return self

    def __next__(self):
        try:
            out = self.models[self.idx]
            self.idx += 1
        except IndexError:
            self.idx = 0
            raise StopIteration
        return out

    def fit(self, model, **kwargs):
        """data.py:385-395.  Emulators exposing `fit_collection` fit every member in one batched GPU call."""
        for process_model in self.models:
            if process_model.distribution != None:  # noqa: E711
                warnings.warn("Removing the model's previously learnt distribution")
        if hasattr(model, "fit_collection"):
            dists = model.fit_collection(self.models, **kwargs)
            for process_model, dist in zip(self.models, dists):
                process_model.distribution = dist
            return
        for process_model in self.models:
            dist = model.fit(process_model, **kwargs)
            process_model.distribution = dist

    def save(self, path: str):
        with open(path, "wb") as f:
            pickle.dump(self, f)

    @property
    def time(self):
        return self.models[0].time

    @property
    def max_val(self):
        return np.max([model.max_val.values for model in self.models])

    @property
    def min_val(self):
        return np.min([model.min_val.values for model in self.models])

    @property
    def number_of_models(self):
        return len(self.models)

    @property
    def model_names(self):
        return [model.model_name for model in self.models]

    def __len__(self):
        return len(self.models)

    def __getitem__(self, item):
        return self.models[item]

    def distributions(self) -> tp.Dict[str, Distribution]:
        return {model.model_name: model.distribution for model in self.models}

    def check_time_axes(self):
        """data.py:542-562"""
        time_axes_match = True
        for model1 in self.models:
            for model2 in self.models:
                t1, t2 = model1.model_data.coords["time"], model2.model_data.coords["time"]
                if t1.shape != t2.shape or np.any(t1 != t2):
                    time_axes_match = False
        if time_axes_match == False:  # noqa: E712
            warnings.warn("Time axes of models don't match: applying naive fix. Check models are collocated "
                          "correctly in time!")
            new_time = self.time
            for model in self.models:
                model.model_data["time"] = new_time
        return
