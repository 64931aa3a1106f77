"""bayesbridge_amd: MI355X-native CG-accelerated coefficient sampler of
bayes-bridge -- a drop-in for the `BayesBridge.gibbs(coef_sampler_type='cg')`
path (reference: OHDSI/bayes-bridge 0.2.6).  The top-level names are the
reference's (bayesbridge/__init__.py:1-4)."""
from ._lib import BbxError, device_count
from .bayesbridge import BayesBridge, SamplerOptions
from .cg_sampler import HipCGSampler
from .device_chain import HipChainBatch, HipGibbsChain
from .design_matrix import (HipDenseDesignMatrix, HipDesignMatrix,
                            HipSparseDesignMatrix)
from .model import LinearModel, LogisticModel, RegressionModel
from .prior import RegressionCoefPrior

__all__ = [
    "BayesBridge", "RegressionModel", "RegressionCoefPrior", "SamplerOptions",
    "HipDesignMatrix", "HipSparseDesignMatrix", "HipDenseDesignMatrix",
    "HipCGSampler", "HipGibbsChain", "HipChainBatch", "LinearModel", "LogisticModel", "BbxError",
    "device_count",
]
