"""CPU oracle for the bayes-bridge CG coefficient sampler.

TEST INFRASTRUCTURE, NOT PRODUCT.  This package restates, in plain NumPy/SciPy,
the reference algorithm of OHDSI/bayes-bridge 0.2.6 for the hot path
(`reg_coef_sampler/cg_sampler.py`, `design_matrix/{sparse,dense}_matrix.py`,
the lines of `reg_coef_sampler.py` and `bayesbridge.py` that call them, and the
two Cython samplers feeding it).  Only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import it, and only as the checker or
as the reported CPU baseline -- never as a fallback of `bayesbridge_amd`.

Parity pin: the restatement is checked in `tests/test_oracle_vs_reference.py`
against (a) the reference's own golden vectors
(`tests/regression_tests/saved_outputs/{linear,logit}_cg_samples.npy`, copied as
data to `tests/golden/`), (b) per-iteration fixtures captured by importing the
reference in the build container (`tests/golden/make_golden.py`), and (c) --
where /root/reference is present -- the imported reference itself.

Third-party arithmetic restated here: SciPy's `scipy.sparse.linalg.cg`
(unpinned by the reference, `setup.py:66-69`; SciPy 1.15.3 semantics, legacy
`tol=` mapped to `rtol=`, `atol=0`) and SciPy's CSR/CSC matvec loops.
"""
from .design_matrix import OracleSparseDesign, OracleDenseDesign, make_design
from .cg_sampler import cg_sample, scipy_style_cg, prior_preconditioner
from .summarizer import CoefSummarizer
