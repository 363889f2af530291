"""Weighting plug-ins with the reference's protocol (ensembles/weights.py:15-53): instantiate, then call with
(process_models, observations=None, **kwargs) -> DataArray with a leading 'model' dimension that sums to one
over 'model'.  The arithmetic of LogLikelihoodWeight (weights.py:56-131) and ModelSimilarityWeight
(weights.py:214-333) runs in libbayens kernels."""
from __future__ import annotations

import abc
import copy

import numpy as np

from . import _lib
from . import distributions as dx
from .data import ModelCollection, ProcessModel
from .labelled import DataArray, ResidentDataArray, concat
from .resident import field_of, resident_tables

__all__ = ["AbstractWeight", "LogLikelihoodWeight", "ModelSimilarityWeight", "InverseSquareWeight", "UniformWeight",
           "KSDWeight", "CRPSWeight"]


class AbstractWeight:
    def __init__(self, name: str) -> None:
        self.name = name

    @abc.abstractmethod
    def _compute(self, process_models: ModelCollection, observations: ProcessModel):
        raise NotImplementedError

    def __call__(self, process_models: ModelCollection, observations: ProcessModel = None, **kwargs):
        """weights.py:27-53"""
        if observations is not None:
            assert np.all(process_models.time.values == observations.time.values), \
                "Time coordinates do not match between models and observations"
            assert len(process_models.time) == len(observations.time), \
                "Time coordinates do not match between models and observations"
        for process_model in process_models.models:
            assert hasattr(process_model.distribution, "_dist"), "Distribution not defined - fit models first"
        return self._compute(process_models=process_models, observations=observations, **kwargs)


class _ResidentTable:
    """Owner of a device table behind a ResidentDataArray: the block goes back to the context when the array is dropped."""

    def __init__(self, ctx, buf):
        self.ctx, self.buf = ctx, buf

    def __del__(self):
        try:
            self.buf.free()
        except Exception:
            pass


def _row_template(model):
    # copy.deepcopy(model.model_data.isel(realisation=0)).drop_vars("realisation")      (weights.py:110-112)
    return copy.deepcopy(model.model_data.isel(realisation=0)).drop_vars("realisation")


class LogLikelihoodWeight(AbstractWeight):
    def __init__(self, name: str = "LogLikelihoodWeight") -> None:
        super().__init__(name)

    def _compute(self, process_models, observations, return_lls=False, standardisation_scheme=None,
                 standardisation_constant=1.0):
        """weights.py:66-131.  standardisation_scheme: None / 'exp' (the reference default jnp.exp) runs fused on
        the device; any other callable is applied on the host to the device-computed mean log-likelihood."""
        ctx = _lib.default_context()
        models = list(process_models.models)
        field = field_of([m.distribution._dist for m in models])
        if field is not None:
            return self._compute_field(field, process_models, observations, models, return_lls, standardisation_scheme,
                                       standardisation_constant)
        obs = np.ascontiguousarray(np.asarray(observations.model_data.values, dtype=np.float64)
                                   .reshape(observations.n_realisations, -1))
        M = len(models)
        P = obs.shape[1]
        ll_mean = np.empty((M, P))
        full = [i for i, m in enumerate(models) if m.distribution.dist_type is not dx.Normal]
        diag = [i for i, m in enumerate(models) if m.distribution.dist_type is dx.Normal]
        if full:
            ds = [models[i].distribution._dist for i in full]
            if not all(isinstance(d, dx.MultivariateNormalFullCovariance) for d in ds):
                for i in full:   # MultivariateNormalDiag etc.: generic log_prob with the (N,1) broadcast
                    d = models[i].distribution._dist
                    ll_mean[i] = np.mean([d.log_prob(o[:, None]) for o in obs], axis=0)
            elif len(full) == M and all(getattr(d, "_resident", None) is not None for d in ds) and \
                    (standardisation_scheme is None or standardisation_scheme == "exp" or standardisation_scheme is np.exp):
                # every posterior is resident in HBM (resident.py): rows, normalisation and the raw likelihoods are
                # computed from the resident basis / log-determinants; only obs goes up and the [M, P] tables come down
                basis, logdet, _, _, _, owned = resident_tables(ctx, ds)
                dobs, dll, dw, draw = ctx.to_device(obs), ctx.empty((M, P)), ctx.empty((M, P)), ctx.empty((M, P))
                try:
                    ctx.loglik_rows_basis_dev(M, P, obs.shape[0], basis, logdet, dobs, dll)
                    ctx.weights_normalise_dev(M, P, dll, standardisation_constant, dw, draw)
                    w, raw = dw.download(), draw.download()
                finally:
                    for d in [dobs, dll, dw, draw] + owned:
                        d.free()
                return self._finish(process_models, observations, models, w, raw, return_lls)
            else:
                basis = np.stack([d._basis for d in ds])
                logdet = np.array([d._logdet for d in ds])
                ll_mean[full] = ctx.loglik_rows_basis(basis, logdet, obs)
        if diag:
            mu = np.stack([models[i].distribution._dist.loc.reshape(-1) for i in diag])
            sc = np.stack([models[i].distribution._dist.scale.reshape(-1) for i in diag])
            ll_mean[diag] = ctx.normal_loglik_rows(mu, sc, obs)

        if standardisation_scheme is None or standardisation_scheme == "exp" or standardisation_scheme is np.exp:
            w, raw = ctx.weights_normalise(ll_mean, standardisation_constant, return_raw=True)
        else:
            raw = np.asarray(standardisation_scheme(standardisation_constant * ll_mean), dtype=np.float64)
            with np.errstate(all="ignore"):
                w = raw / raw.sum(axis=0, keepdims=True)

        return self._finish(process_models, observations, models, w, raw, return_lls)

    def _compute_field(self, field, process_models, observations, models, return_lls, scheme, c_std):
        """Per-gridpoint posteriors (GPDTW1D(per_gridpoint=True), BASELINE config 5): weights.py:88-123 at EVERY grid
        point - rows log N(x_{r,t,g} 1; mu_{m,g}, cov_{m,g}) from the resident basis of fit (m, g), mean over the
        observation realisations, exp(c .), normalisation over the models - in three launches over the whole field
        (bayens_loglik_rows_basis_grouped -> bayens_weights_normalise).  Only the observations go up; the
        [model, time, *space] weight table STAYS in HBM (round 4: labelled.ResidentDataArray - `Barycentre` reads it where it
        lies; `.values`, arithmetic or a reduction download it once).  A standardisation scheme other than exp is a host
        callable (weights.py:66-73,107 takes any): the mean log-likelihood table comes down and the callable is applied to
        it as the 1-D path does."""
        ctx = field.ctx
        M, G, N = field.M, field.G, field.N
        R_obs = observations.n_realisations
        on_device = scheme is None or scheme == "exp" or scheme is np.exp
        # observations go up as they lie in memory [R_obs][N][G] and are re-laid on the device to [G][R_obs][N] (one
        # observation set per grid point): [R_obs * N][G] -> [G][R_obs * N] is that, since rows of one point stay together
        dobs_raw = ctx.to_device(np.asarray(observations.model_data.values, dtype=np.float64).reshape(R_obs * N, G))
        dobs, dll, dw = ctx.empty((G, R_obs, N)), ctx.empty((M * G, N)), ctx.empty((M, G * N))
        draw = ctx.empty((M, G * N)) if return_lls and on_device else None
        keep = []
        try:
            ctx.transpose_rows_dev(1, R_obs * N, G, dobs_raw, dobs)
            ctx.loglik_rows_basis_grouped(field.basis, field.logdet, dobs, loc=_lib.DEVICE, out=dll)
            if not on_device:
                dout = ctx.empty((M, N * G))
                try:
                    ctx.transpose_rows_dev(M, G, N, dll, dout)              # device order [M][G][N] -> (model, time, *space)
                    ll_mean = dout.download()
                finally:
                    dout.free()
                raw = np.asarray(scheme(c_std * ll_mean), dtype=np.float64)
                with np.errstate(all="ignore"):
                    w = raw / raw.sum(axis=0, keepdims=True)
                return self._finish(process_models, observations, models, w, raw, return_lls)
            ctx.weights_normalise_dev(M, G * N, dll, c_std, dw, draw)
            keep = [dw] + ([draw] if draw is not None else [])
        finally:
            for d in (dobs_raw, dobs, dll, dw, draw):
                if d is not None and not any(d is k for k in keep):
                    d.free()
        template = models[0].model_data
        dims = ("model",) + tuple(template.dims[1:])
        coords = {k: v for k, v in template.coords.items() if k in dims}
        coords["model"] = np.asarray([m.model_name for m in models])
        shape = (M,) + tuple(template.shape[1:])

        def lazy(table, name):
            owner = _ResidentTable(ctx, table)

            def fetch():
                dout = ctx.empty((M, N * G))
                try:
                    ctx.transpose_rows_dev(M, G, N, owner.buf, dout)        # device order [M][G][N] -> the data's (time, *space)
                    return dout.download()
                finally:
                    dout.free()
            return ResidentDataArray(fetch, shape, dims, coords, name, resident=(owner, field, "member-gridpoint-time"))

        weights = lazy(dw, "Log-likelihood weights")
        assert weights.shape == (len(process_models),) + tuple(observations.model_data.shape[1:])
        if return_lls:
            return weights, lazy(draw, "Log-likelihoods")
        return weights

    @staticmethod
    def _finish(process_models, observations, models, w, raw, return_lls):
        def to_da(vals, name):
            rows = []
            for i, m in enumerate(models):
                row = _row_template(m)
                row.data = vals[i].reshape(row.shape)
                rows.append(row.assign_coords(model=m.model_name))
            return concat(rows, dim="model").rename(name)

        weights = to_da(w, "Log-likelihood weights")
        obs_real_shape = observations.model_data.shape[1:]
        assert weights.shape == (len(process_models),) + obs_real_shape
        if return_lls:
            return weights, to_da(raw, "Log-likelihoods")
        return weights


class ModelSimilarityWeight(AbstractWeight):
    def __init__(self, name: str = "ModelSimilarityWeight") -> None:
        super().__init__(name)

    def _compute(self, process_models, mode: str = "single", observations: ProcessModel = None):
        """weights.py:224-333.  'temporal' runs on the device; `observations` is accepted and ignored as in the
        reference."""
        if mode == "temporal":
            ctx = _lib.default_context()
            ds = [m.distribution._dist for m in process_models.models]
            if all(getattr(d, "_resident", None) is not None for d in ds):
                _, _, mean, var, _, owned = resident_tables(ctx, ds)     # marginals straight from HBM
                M, P = len(ds), ds[0]._resident[0].N
                dw = ctx.empty((M, P))
                try:
                    ctx.w2_temporal_dev(M, P, mean, var, dw)
                    w = dw.download()
                finally:
                    for d in [dw] + owned:
                        d.free()
            else:
                mean = np.stack([np.asarray(d.mean()).reshape(-1) for d in ds])
                var = np.stack([np.asarray(d.variance()).reshape(-1) for d in ds])
                w = ctx.w2_temporal(mean, var)
            out = DataArray(w, ("model", "time"),
                            {"model": np.asarray(process_models.model_names),
                             "time": process_models[0].model_data.coords["time"]},
                            "Model similarity weights")
            return out
        if mode == "single":
            ctx = _lib.default_context()
            dists = [m.distribution._dist for m in process_models.models]
            if all(getattr(d, "_resident", None) is not None for d in dists):
                _, _, mean, _, chol, owned = resident_tables(ctx, dists, want_chol=True)
                if chol is not None:                       # fp64 factors: they never leave HBM
                    M, N = len(dists), dists[0]._resident[0].N
                    dw, dD = ctx.empty((M,)), ctx.empty((M, M))
                    try:
                        ctx.w2_single_dev(M, N, mean, chol, dw, dD)
                        w = dw.download()
                    finally:
                        for d in [dw, dD] + owned:
                            d.free()
                    return DataArray(w[:, None], ("model", "time"),
                                     {"model": np.asarray(process_models.model_names), "time": np.asarray([0])},
                                     "Model similarity weights")
                for d in owned:
                    d.free()
            mu = np.stack([np.asarray(d.mean()).reshape(-1) for d in dists])
            if all(isinstance(d, dx.MultivariateNormalFullCovariance) for d in dists):
                w, _ = ctx.w2_single(mu, chol=np.stack([d._scale_tri for d in dists]))
            elif all(isinstance(d, dx.Normal) for d in dists):
                # full_cov=False: sigma = diag(variance()) (wasserstein.py:35-38) -> the trace term is sum (sd_i - sd_j)^2
                w, _ = ctx.w2_single(mu, sd=np.stack([np.asarray(d.stddev()).reshape(-1) for d in dists]))
            else:
                raise NotImplementedError("mode='single' needs all-full-covariance or all-dx.Normal posteriors")
            return DataArray(w[:, None], ("model", "time"),
                             {"model": np.asarray(process_models.model_names), "time": np.asarray([0])},
                             "Model similarity weights")
        if mode == "spatial":
            # weights.py:267-300.  Only the dx.Normal branch exists in the reference: its other branch builds
            # `dx.MultiVariate`, which distrax does not have (weights.py:284 - AttributeError at the first grid point).
            import warnings
            warnings.warn("Spatial method is experimental. Use with caution.")
            models = list(process_models.models)
            if not all(m.distribution.dist_type is dx.Normal for m in models):
                raise NotImplementedError(
                    "ModelSimilarityWeight(mode='spatial') is defined for dx.Normal posteriors (MeanFieldApproximation); "
                    "the reference's branch for other posteriors calls a non-existent dx.MultiVariate (weights.py:284)")
            md = models[0].model_data
            if md.ndim != 4:
                raise ValueError("mode='spatial' needs (realisation, time, latitude, longitude) models")
            T, n_lat, n_lon = md.shape[1:]
            mean = np.stack([np.asarray(m.distribution.mean.values, dtype=np.float64).reshape(T, -1) for m in models])
            V = np.stack([np.asarray(m.distribution.variance.values, dtype=np.float64).reshape(T, -1) for m in models])
            w = _lib.default_context().w2_spatial(mean, V).reshape(len(models), n_lat, n_lon)
            lat_name, lon_name = md.dims[2], md.dims[3]
            return DataArray(w, ("model", lat_name, lon_name),
                             {"model": np.asarray(process_models.model_names), lat_name: md.coords[lat_name],
                              lon_name: md.coords[lon_name]}, "Model similarity weights")
        raise ValueError('Mode must be "single", "spatial", or "temporal"')


def _packed_realisations(models):
    """[M][R_max][P] zero-padded realisations + the realisation counts (members may differ in R)."""
    R = np.array([m.n_realisations for m in models], dtype=np.int32)
    P = int(models[0].model_data.size / models[0].n_realisations)
    data = np.zeros((len(models), int(R.max()), P))
    for i, m in enumerate(models):
        data[i, :R[i]] = np.asarray(m.model_data.values, dtype=np.float64).reshape(R[i], -1)
    return data, R


def _stack_rows(models, vals, name):
    rows = []
    for i, m in enumerate(models):
        row = _row_template(m)
        row.data = vals[i].reshape(row.shape)
        rows.append(row.assign_coords(model=m.model_name))
    return concat(rows, dim="model").rename(name)


class InverseSquareWeight(AbstractWeight):
    def __init__(self, name: str = "InverseSquareWeight") -> None:
        super().__init__(name)

    def _compute(self, process_models, observations):
        """weights.py:147-174: (mean_r model - mean_r obs)^-2, normalised over the models."""
        models = list(process_models.models)
        data, R = _packed_realisations(models)
        obs = np.asarray(observations.model_data.values, dtype=np.float64).reshape(observations.n_realisations, -1)
        w = _lib.default_context().inverse_square_weights(data, R, obs)
        weights = _stack_rows(models, w, "Inverse square weights")
        assert weights.shape[1] == models[-1].model_data.shape[1], \
            "Weight is not the same size as model. Check observations and model time coordinates match!"
        return weights


class UniformWeight(AbstractWeight):
    def __init__(self, name: str = "UniformWeight") -> None:
        super().__init__(name)

    def _compute(self, process_models, observations):
        """weights.py:190-212: 1 / n_models everywhere (exp(0) normalised over the models on the device)."""
        models = list(process_models.models)
        P = int(models[0].model_data.size / models[0].n_realisations)
        w = _lib.default_context().weights_normalise(np.zeros((len(models), P)), 1.0)
        return _stack_rows(models, w, "Uniform weights")


class _ScoreWeight(AbstractWeight):
    kind = None
    label = None

    def _compute(self, process_models, observations):
        assert len(process_models.time) == len(observations.time), \
            "Time coordinates do not match between models and observations"
        assert hasattr(process_models[0].distribution, "_dist"), "Distribution not defined - fit models first"
        models = list(process_models.models)
        obs = np.asarray(observations.model_data.values, dtype=np.float64).reshape(observations.n_realisations, -1)
        mean = np.stack([np.asarray(m.distribution._dist.mean()).reshape(-1) for m in models])
        var = np.stack([np.asarray(m.distribution._dist.variance()).reshape(-1) for m in models])
        w, _ = _lib.default_context().score_weights(self.kind, mean, var, obs)
        return _stack_rows(models, w, self.label)


class KSDWeight(_ScoreWeight):
    """weights.py:336-441: weight ~ 1 / (IMQ kernel Stein discrepancy of the observation values at each point
    against dx.Normal(mean_i, scale = variance_i))."""
    kind, label = "ksd", "Kernel Stein Discrepancy weights"

    def __init__(self, name: str = "KernelSteinDiscrepancyWeight") -> None:
        super().__init__(name)


class CRPSWeight(_ScoreWeight):
    """weights.py:444-515: weight ~ 1 / mean_r crps_gaussian(obs_r; mean_i, sig = variance_i) - the scheme the
    reference's published experiment uses (experiments/full_experiment_script.py:116)."""
    kind, label = "crps", "Continuous Ranked Probability Scores weights"

    def __init__(self, name: str = "ContinuousRankedProbabilityScoreWeight") -> None:
        super().__init__(name)
