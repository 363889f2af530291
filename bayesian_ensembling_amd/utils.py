"""PerfectModelTest (ensembles/utils.py:32-225): leave-one-model-out test of the fit -> weight -> ensemble chain
with the reference's three metrics (NLL, RMSE, W2) for the barycentre and for the multi-model mean.  Same
constructor and `run` signature; figures are not produced (plotting is out of scope), the CSV is written when
`save_dir` is given.  `load_model_collection(path)` mirrors utils.py:22-30 (without its stray `self`)."""
from __future__ import annotations

import copy
import csv
import os
import pickle

import numpy as np

from . import distributions as dx
from .data import ModelCollection
from .wasserstein import gaussian_w2_distance_distrax
from .weights import ModelSimilarityWeight


def load_model_collection(path):
    with open(path, "rb") as f:
        return pickle.load(f)


class PerfectModelTest:
    def __init__(self, hindcast_models, forecast_models, emulate_method, weight_method, ensemble_method, ssp: str,
                 include_sim: bool = False, save_dir: str = None):
        self.hindcast_models = hindcast_models
        self.forecast_models = forecast_models
        self.emulate_method = emulate_method
        self.weight_method = weight_method
        self.ensemble_method = ensemble_method
        self.ssp = ssp
        self.save_dir = save_dir
        self.include_sim = include_sim
        self.save_csv_dir = None
        if save_dir is not None:
            self.save_csv_dir = os.path.join(save_dir, "csvs")
            os.makedirs(self.save_csv_dir, exist_ok=True)

    def _run_single_test(self, hindcast_models, forecast_models, pseudo_observations_past, pseudo_observations_future,
                         n_optim_nits: int = 1000, use_prefit_models: bool = False):
        """utils.py:83-181 without the figures."""
        if use_prefit_models != True:  # noqa: E712
            hindcast_models.fit(model=self.emulate_method(), compile_objective=True, n_optim_nits=n_optim_nits,
                                progress_bar=False)
            forecast_models.fit(model=self.emulate_method(), compile_objective=True, n_optim_nits=n_optim_nits,
                                progress_bar=False)
            pseudo_observations_future.distribution = self.emulate_method().fit(
                pseudo_observations_future, compile_objective=True, n_optim_nits=n_optim_nits)

        weight_function = self.weight_method()
        weights = weight_function(hindcast_models, pseudo_observations_past)
        mean_weights = weights.mean("time")
        if self.include_sim:
            sim_weights = ModelSimilarityWeight()(hindcast_models, pseudo_observations_future)
            total_weights = mean_weights * sim_weights.mean("time")
            total_weights = total_weights / total_weights.sum()
        else:
            total_weights = mean_weights
        weights_single = total_weights.expand_dims(time=forecast_models[0].model_data.coords["time"], axis=1)
        barycentre = self.ensemble_method()(forecast_models, weights_single)

        obs = pseudo_observations_future.model_data
        nll_bary = -float(np.mean(barycentre._dist.log_prob(obs.values)))
        # (time) - (realisation, time) -> dims (time, realisation): axis 0 is time            (utils.py:141)
        rmse_bary = float(np.mean(np.sqrt(np.mean(((barycentre.mean - obs) ** 2).values, axis=0))))
        full_cov = hasattr(pseudo_observations_future.distribution._dist, "covariance")
        w2_bary = gaussian_w2_distance_distrax(barycentre._dist, pseudo_observations_future.distribution._dist,
                                               full_cov=full_cov)

        realisations = np.vstack([m.model_data.values for m in forecast_models.models])
        mmm_dist = dx.Normal(np.mean(realisations, axis=0), np.var(realisations, axis=0))
        nll_mmm = -float(np.mean(mmm_dist.log_prob(obs.values)))
        # ndarray - (realisation, time) keeps (realisation, time): axis 0 is the realisation    (utils.py:151)
        rmse_mmm = float(np.mean(np.sqrt(np.mean((mmm_dist.mean()[None, :] - obs.values) ** 2, axis=0))))
        w2_mmm = gaussian_w2_distance_distrax(mmm_dist, pseudo_observations_future.distribution._dist, full_cov=False)
        return nll_bary, rmse_bary, w2_bary, nll_mmm, rmse_mmm, w2_mmm

    def run(self, n_optim_nits: int = 1000, use_prefit_models=False):
        """utils.py:184-225.  Returns the rows (also written as CSV when save_dir was given)."""
        name = self.weight_method().name
        header = ["model as psuedo obs", f"nll_bary_{name}", f"rmse_bary_{name}", f"w2_bary_{name}", "nll_mmm",
                  "rmse_mmm", "w2_mmm"]
        rows = []
        for i in range(self.hindcast_models.number_of_models):
            hindcast_model_list = copy.deepcopy(self.hindcast_models.models)
            pseudo_observations_past = hindcast_model_list.pop(i)
            forecast_model_list = copy.deepcopy(self.forecast_models.models)
            pseudo_observations_future = forecast_model_list.pop(i)
            res = self._run_single_test(ModelCollection(hindcast_model_list), ModelCollection(forecast_model_list),
                                        pseudo_observations_past, pseudo_observations_future, n_optim_nits,
                                        use_prefit_models=use_prefit_models)
            rows.append([pseudo_observations_past.model_name, *res])
        if self.save_csv_dir:
            suffix = f"{name}_plus_sim_{self.ssp}" if self.include_sim else f"{name}_{self.ssp}"
            save_file = os.path.join(self.save_csv_dir, f"prefect_model_test_results_{suffix}.csv")
            with open(save_file, "w", newline="") as f:
                w = csv.writer(f)
                w.writerow([""] + header)
                for k, r in enumerate(rows):
                    w.writerow([k] + r)
            print(f"Saved results to {save_file}")
        return header, rows
