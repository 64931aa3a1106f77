"""Import the upstream reference (OHDSI/bayes-bridge) as a parity oracle.

TEST INFRASTRUCTURE ONLY, and only usable in the build container: the
reference lives read-only under /root/reference and never travels to the GPU
box.  This helper

  1. copies /root/reference to a scratch directory under /tmp and runs the
     reference's own ``setup.py build_ext --inplace`` there (4 Cython RNG
     extensions; nothing is written to /root/reference or to this repo);
  2. installs two in-process compatibility shims (no reference file edited):
       * ``scipy.sparse.linalg.cg`` accepts the legacy ``tol=`` keyword that
         the reference passes (cg_sampler.py:77-80) and maps it to
         ``rtol=tol, atol=0`` -- SciPy >= 1.14 removed ``tol``;
       * ``np.int = int`` for cox_model.py:156,166,175 (off the hot path; only
         needed so ``import bayesbridge`` succeeds on NumPy >= 1.24);
  3. puts the scratch copy on ``sys.path`` and imports ``bayesbridge`` and the
     reference's ``simulate_data`` module.

Used by ``make_golden.py`` (fixture generation) and by the ``needs_reference``
tests, which are skipped wherever /root/reference is absent.
"""
import os
import shutil
import subprocess
import sys

REFERENCE_ROOT = os.environ.get("BBX_REFERENCE_ROOT", "/root/reference")
SCRATCH = os.environ.get("BBX_REFERENCE_SCRATCH", "/tmp/bbx_refbuild")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "bayesbridge"))


def _build_scratch():
    marker = os.path.join(SCRATCH, ".built")
    if os.path.exists(marker):
        return
    if os.path.exists(SCRATCH):
        shutil.rmtree(SCRATCH)
    shutil.copytree(REFERENCE_ROOT, SCRATCH)
    for root, dirs, files in os.walk(SCRATCH):
        for name in dirs + files:
            path = os.path.join(root, name)
            os.chmod(path, os.stat(path).st_mode | 0o200)
    os.chmod(SCRATCH, os.stat(SCRATCH).st_mode | 0o200)
    log = os.path.join(SCRATCH, "build.log")
    with open(log, "w") as fh:
        subprocess.check_call(
            [sys.executable, "setup.py", "build_ext", "--inplace"],
            cwd=SCRATCH, stdout=fh, stderr=subprocess.STDOUT)
    open(marker, "w").close()


def _install_shims():
    import numpy as np
    import scipy.sparse.linalg as spla
    if not hasattr(np, "int"):
        np.int = int
    if getattr(spla.cg, "_bbx_tol_shim", False):
        return
    modern_cg = spla.cg

    def cg_with_legacy_tol(A, b, x0=None, *, tol=None, rtol=1e-5, atol=0.,
                           maxiter=None, M=None, callback=None):
        if tol is not None:
            rtol, atol = tol, 0.
        return modern_cg(A, b, x0=x0, rtol=rtol, atol=atol, maxiter=maxiter,
                         M=M, callback=callback)

    cg_with_legacy_tol._bbx_tol_shim = True
    spla.cg = cg_with_legacy_tol
    import scipy.sparse
    scipy.sparse.linalg.cg = cg_with_legacy_tol


def import_reference():
    """Returns (bayesbridge module, simulate_data module) of the reference."""
    if not reference_available():
        raise RuntimeError("reference not present at " + REFERENCE_ROOT)
    _build_scratch()
    _install_shims()
    if SCRATCH not in sys.path:
        sys.path.insert(0, SCRATCH)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import bayesbridge
        import simulate_data
    return bayesbridge, simulate_data
