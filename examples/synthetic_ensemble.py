"""End-to-end use of the drop-in API on a synthetic ensemble (needs an MI355X):
fit every member's GP posterior in one batched call, weight the members against observations, combine them.

    python examples/synthetic_ensemble.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesian_ensembling_amd as es                                  # noqa: E402
from bayesian_ensembling_amd.labelled import DataArray               # noqa: E402
from bayesian_ensembling_amd.synthetic import synthetic_ensemble     # noqa: E402
from bayesian_ensembling_amd.weights import *                        # noqa: E402,F401,F403


def main(n_models=6, n_time=165, n_real=5):
    data, obs = synthetic_ensemble(n_models, n_time, R=n_real, R_obs=3)
    time = np.arange(1850, 1850 + n_time)

    def pm(values, name):
        da = DataArray(values, ("realisation", "time"), {"realisation": np.arange(values.shape[0]), "time": time})
        return es.ProcessModel(da, name)

    models = es.ModelCollection([pm(data[i], f"model{i}") for i in range(n_models)])
    observations = pm(obs, "obs")

    models.fit(model=es.GPDTW1D(target="dba"), n_optim_nits=50, compile_objective=True, progress_bar=False)
    for m in models:
        h = m.distribution.hyperparameters
        print(f"{m.model_name}: kernel variance {h['variance']:.3f}, lengthscale {h['lengthscale']:.3f}")

    ll_w = LogLikelihoodWeight()(models, observations, standardisation_constant=1e-3)
    crps_w = CRPSWeight()(models, observations)
    sim_w = ModelSimilarityWeight()(models)                       # full-covariance Wasserstein similarity
    print("time-mean log-likelihood weights:", np.round(ll_w.mean("time").values, 3))
    print("time-mean CRPS weights:          ", np.round(crps_w.mean("time").values, 3))
    print("model-similarity weights:        ", np.round(sim_w.values[:, 0], 3))

    post = es.Barycentre()(models, crps_w)
    mmm = es.MultiModelMean()(models, None)
    print("barycentre mean at the last step %.3f (sd %.3f); multi-model mean %.3f"
          % (post.mean.values[-1], np.sqrt(post.covariance[-1]), mmm.mean.values[-1]))


if __name__ == "__main__":
    main()
