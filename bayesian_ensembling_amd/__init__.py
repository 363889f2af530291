"""bayesian_ensembling_amd - MI355X-native drop-in for the GP-fit -> ensemble-weighting -> weighted-posterior
path of mattramos/bayesian_ensembling (reference package name: `ensembles`).

Same public names as ensembles/__init__.py for that path; the arithmetic runs in hand-written HIP kernels
(libbayens.so, include/bayens.h) reached through ctypes.  There is no CPU fallback."""
from .data import Distribution, ModelCollection, ProcessModel
from .dtwa import performDBA
from .ensemble_scheme import AbstractEnsembleScheme, Barycentre, MultiModelMean, WeightedModelMean
from .labelled import DataArray, concat
from .models import GPDTW1D, MeanFieldApproximation, fit_collections
from .nc4 import open_dataarray
from .weights import (AbstractWeight, CRPSWeight, InverseSquareWeight, KSDWeight, LogLikelihoodWeight,
                      ModelSimilarityWeight, UniformWeight)
from . import distributions

__version__ = "0.1.0"
