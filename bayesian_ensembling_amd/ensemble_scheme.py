"""Ensembling plug-ins (ensembles/ensemble_scheme.py): Scheme()(process_models, weights, **kw) -> Distribution."""
from __future__ import annotations

import abc
import typing as tp

import numpy as np

from . import _lib
from . import distributions as dx
from .data import Distribution, ModelCollection
from .labelled import DataArray, ones_like
from .resident import field_of, resident_tables


class AbstractEnsembleScheme:
    def __init__(self, name: str) -> None:
        self.name = name
        self.distributions = None

    @abc.abstractmethod
    def _compute(self, process_models: ModelCollection, weights) -> Distribution:
        raise NotImplementedError

    def __call__(self, process_models: ModelCollection, weights, **kwargs) -> tp.Any:
        return self._compute(process_models=process_models, weights=weights, **kwargs)   # ensemble_scheme.py:33-36


def _blank(process_model):
    return (ones_like(process_model.model_data[0].drop_vars("realisation")) * np.nan).rename("blank")


class Barycentre(AbstractEnsembleScheme):
    def __init__(self, name: str = "Barycentre") -> None:
        super().__init__(name)

    def _compute(self, process_models, weights, n_threads=2) -> Distribution:
        """ensemble_scheme.py:43-81: per point the 1-D Wasserstein barycentre of the members' marginals
        N(mean_t, variance_t).  Returns MultivariateNormalDiag(mu*, scale_diag = sigma*^2) as the reference does."""
        del n_threads
        M = process_models.number_of_models
        for t_idx, process_model in enumerate(process_models.models):
            if not process_model.distribution:
                raise AttributeError(f"No posterior for model {t_idx}. Please run model.fit() first.")
        n_points = int(process_models[0].model_data.size / process_models[0].model_data.shape[0])
        ds = [pm.distribution._dist for pm in process_models.models]
        field = field_of(ds)
        res = getattr(weights, "resident", None)
        if field is not None and res is not None and res[1] is field and res[2] == "member-gridpoint-time" \
                and tuple(weights.shape) == (M,) + tuple(process_models[0].model_data.shape[1:]):
            # the weight table is still where LogLikelihoodWeight left it, in the field's own [member][grid point][time]
            # layout: no download, no upload, no change of layout (ensemble_scheme.py:52 reshapes the same table on the host)
            ctx, G, N = field.ctx, field.G, field.N
            dmu, dv, dout = ctx.empty((G * N,)), ctx.empty((G * N,)), ctx.empty((2, N * G))
            try:
                bad = ctx.barycentre_dev(M, G * N, res[0].buf, field.mu, field.var, dmu, dv)
                ctx.transpose_rows_dev(1, G, N, dmu, dout)
                ctx.transpose_rows_dev(1, G, N, dv, dout.offset(N * G * 8))
                both = dout.download()
                mu, v = both[0], both[1]
            finally:
                for d in (dmu, dv, dout):
                    d.free()
            if bad:
                import warnings
                warnings.warn(f"Barycentre not converged for {bad} time step(s)")
            return Distribution(mu=mu, covariance=v, dim_array=_blank(process_models.models[-1]),
                                dist_type=dx.MultivariateNormalDiag)
        w = np.asarray(getattr(weights, "values", weights), dtype=np.float64).reshape(M, -1)
        if w.shape[1] != n_points:
            raise ValueError(f"weights have {w.shape[1]} points per model, the posteriors {n_points}")
        if field is not None:
            # per-gridpoint posteriors: the marginals of all M x G fits are resident as [M][G*N]; the weights arrive in
            # the data's (time, *space) order and are turned into that layout, the result back (ensemble_scheme.py:51-52
            # flattens space into the point axis in the same way)
            ctx, G, N = field.ctx, field.G, field.N
            dwt, dw, dmu, dv, dout = ctx.to_device(w), ctx.empty((M, G * N)), ctx.empty((G * N,)), ctx.empty((G * N,)), ctx.empty((2, N * G))
            try:
                ctx.transpose_rows_dev(M, N, G, dwt, dw)                               # (time, space) -> [grid point][time]
                bad = ctx.barycentre_dev(M, G * N, dw, field.mu, field.var, dmu, dv)
                ctx.transpose_rows_dev(1, G, N, dmu, dout)
                ctx.transpose_rows_dev(1, G, N, dv, dout.offset(N * G * 8))
                both = dout.download()
                mu, v = both[0], both[1]
            finally:
                for d in (dwt, dw, dmu, dv, dout):
                    d.free()
        elif all(getattr(d, "_resident", None) is not None for d in ds):
            # resident posteriors: the [M, P] marginals are read in HBM; the weights go up, mu* / sigma*^2 come down
            ctx = _lib.default_context()
            _, _, mean, var, _, owned = resident_tables(ctx, ds)
            dw, dmu, dv = ctx.to_device(w), ctx.empty((n_points,)), ctx.empty((n_points,))
            try:
                bad = ctx.barycentre_dev(M, n_points, dw, mean, var, dmu, dv)
                mu, v = dmu.download(), dv.download()
            finally:
                for d in [dw, dmu, dv] + owned:
                    d.free()
        else:
            mean = np.stack([np.asarray(d.mean()).reshape(-1) for d in ds])
            var = np.stack([np.asarray(d.variance()).reshape(-1) for d in ds])
            mu, v, bad = _lib.default_context().barycentre(w, mean, var)
        if bad:
            import warnings
            warnings.warn(f"Barycentre not converged for {bad} time step(s)")
        return Distribution(mu=mu, covariance=v, dim_array=_blank(process_models.models[-1]),
                            dist_type=dx.MultivariateNormalDiag)


class MultiModelMean(AbstractEnsembleScheme):
    def __init__(self, name: str = "MultiModelMean") -> None:
        super().__init__(name)

    def _compute(self, process_models, weights=None) -> Distribution:
        """ensemble_scheme.py:108-128: mean and std**2 over every realisation of every model."""
        models = list(process_models.models)
        R = np.array([m.n_realisations for m in models], dtype=np.int32)
        P = int(models[0].model_data.size / models[0].n_realisations)
        data = np.zeros((len(models), int(R.max()), P))
        for i, m in enumerate(models):
            data[i, :R[i]] = np.asarray(m.model_data.values, dtype=np.float64).reshape(R[i], -1)
        mu, v = _lib.default_context().multi_model_mean(data, R)
        return Distribution(mu=mu, covariance=v, dim_array=_blank(models[0]), dist_type=dx.MultivariateNormalDiag)


class WeightedModelMean(AbstractEnsembleScheme):
    def __init__(self, name: str = "MultiModelMean") -> None:   # (sic) the reference names it "MultiModelMean"
        super().__init__(name)

    def _compute(self, process_models, weights) -> Distribution:
        """ensemble_scheme.py:136-160"""
        models = list(process_models.models)
        M = len(models)
        wda = DataArray.from_any(weights)
        w = np.stack([np.asarray(wda.sel(model=m.model_name).values, dtype=np.float64).reshape(-1) for m in models])
        R = np.array([m.n_realisations for m in models], dtype=np.int32)
        P = int(models[0].model_data.size / models[0].n_realisations)
        if w.shape[1] == 1 and P > 1:
            w = np.repeat(w, P, axis=1)
        data = np.zeros((M, int(R.max()), P))
        for i, m in enumerate(models):
            data[i, :R[i]] = np.asarray(m.model_data.values, dtype=np.float64).reshape(R[i], -1)
        mu, v = _lib.default_context().weighted_model_mean(data, R, w)
        return Distribution(mu=mu, covariance=v, dim_array=_blank(models[0]), dist_type=dx.MultivariateNormalDiag)
