"""bayesbridge_amd: MI355X-native CG-accelerated coefficient sampler of
bayes-bridge (drop-in for the `coef_sampler_type='cg'` path)."""
from .design_matrix import (HipDesignMatrix, HipSparseDesignMatrix,
                            HipDenseDesignMatrix)
from .cg_sampler import HipCGSampler
from ._lib import BbxError, device_count

__all__ = [
    "HipDesignMatrix", "HipSparseDesignMatrix", "HipDenseDesignMatrix",
    "HipCGSampler", "BbxError", "device_count",
]
