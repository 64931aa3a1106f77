"""ctypes binding of libbbx.so (the C ABI declared in include/bbx.h).

The reference's precedent for this layer is design_matrix/mkl_matvec.py:1-56
(ctypes -> MKL).  There is NO CPU fallback: if the HIP library is missing or no
MI355X is visible, the product path raises.
"""
import ctypes
import os
from ctypes import (POINTER, byref, c_char_p, c_double, c_int, c_int32,
                    c_int64, c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libbbx.so")

ABI_VERSION = 103          # BBX_VERSION of include/bbx.h
FORMAT_AUTO, FORMAT_CSR, FORMAT_TILED = 0, 1, 2
F64, F32 = 0, 1
MODEL_LINEAR, MODEL_LOGIT = 0, 1
GSCALE_SAMPLE, GSCALE_OPTIMIZE, GSCALE_FIXED = 0, 1, 2
BATCH_ALLOW_SLOW = 1
# Philox stream ids of the chain's draws (csrc/philox.hpp)
STREAM_ETA1, STREAM_ETA2 = 1, 2

_lib = None
# Set by an atexit hook: objects that are only collected while the interpreter
# shuts down (frames kept alive by a traceback, module globals) no longer call
# into the library -- the order in which the HIP runtime, torch and those
# objects go away is not ours to choose (seen as a segfault at exit after a
# failed test); the process is ending, the driver reclaims the memory.
finalizing = False


def _mark_finalizing():
    global finalizing
    finalizing = True


import atexit  # noqa: E402
atexit.register(_mark_finalizing)


class BbxError(RuntimeError):
    """A libbbx call returned a negative status."""


def _declare(lib):
    dp = POINTER(c_double)
    ip = POINTER(c_int32)
    hp = c_void_p
    sigs = {
        "bbx_version": ([], c_int),
        "bbx_last_error": ([], c_char_p),
        "bbx_device_count": ([POINTER(c_int)], c_int),
        "bbx_builder_threads": ([POINTER(c_int)], c_int),
        "bbx_setup_lock_acquire": ([], c_int),
        "bbx_setup_lock_release": ([], c_int),
        "bbx_design_create_csr": (
            [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
             c_int, c_int, c_int, POINTER(hp)], c_int),
        "bbx_design_create_csr64": (
            [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
             c_int, c_int, c_int, POINTER(hp)], c_int),
        "bbx_design_create_csr_dev": (
            [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
             c_int, c_int, c_int, POINTER(hp)], c_int),
        "bbx_design_create_dense": (
            [c_int64, c_int64, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
             POINTER(hp)], c_int),
        "bbx_design_create_dense_dev": (
            [c_int64, c_int64, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
             POINTER(hp)], c_int),
        "bbx_design_destroy": ([hp], c_int),
        "bbx_design_shape": ([hp, POINTER(c_int64), POINTER(c_int64)], c_int),
        "bbx_design_nnz": ([hp, POINTER(c_int64)], c_int),
        "bbx_design_is_sparse": ([hp, POINTER(c_int)], c_int),
        "bbx_design_is_binary": ([hp, POINTER(c_int)], c_int),
        "bbx_design_device": ([hp, POINTER(c_int)], c_int),
        "bbx_design_format": ([hp, POINTER(c_int)], c_int),
        "bbx_design_storage_bytes": ([hp, POINTER(c_int64)], c_int),
        "bbx_design_matvec_bytes": (
            [hp, POINTER(c_int64), POINTER(c_int64)], c_int),
        "bbx_design_useful_bytes": (
            [hp, POINTER(c_int64), POINTER(c_int64), POINTER(c_double),
             POINTER(c_double)], c_int),
        "bbx_design_timed_bytes": (
            [hp, POINTER(c_int64), POINTER(c_int64)], c_int),
        "bbx_design_fused_operator_bytes": ([hp, POINTER(c_int64)], c_int),
        "bbx_design_cg_launches": ([hp, POINTER(c_int)], c_int),
        "bbx_design_cg_stats": ([hp, POINTER(c_int64), POINTER(c_int64),
                                 POINTER(c_int64), c_int], c_int),
        "bbx_launch_count": ([], c_uint64),
        "bbx_design_set_cg_fold": ([hp, c_int], c_int),
        "bbx_design_hybrid_info": (
            [hp, POINTER(c_int), POINTER(c_int64), POINTER(c_int64),
             POINTER(c_int64), POINTER(c_int)], c_int),
        "bbx_design_tiled_info": (
            [hp, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int),
             POINTER(c_int), POINTER(c_int64), POINTER(c_int64),
             POINTER(c_int)], c_int),
        "bbx_design_dot": ([hp, c_void_p, c_void_p], c_int),
        "bbx_design_tdot": ([hp, c_void_p, c_void_p], c_int),
        "bbx_design_dot_dev": ([hp, c_void_p, c_void_p], c_int),
        "bbx_design_tdot_dev": ([hp, c_void_p, c_void_p], c_int),
        "bbx_design_gram_matvec": ([hp, c_void_p, c_void_p, c_void_p], c_int),
        "bbx_design_gram_matvec_dev": (
            [hp, c_void_p, c_void_p, c_void_p], c_int),
        "bbx_design_stream": ([hp, POINTER(c_void_p)], c_int),
        "bbx_design_synchronize": ([hp], c_int),
        "bbx_cg_sample": (
            [hp, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
             c_void_p, c_void_p, c_uint64, c_int, c_double, c_void_p,
             POINTER(c_int), POINTER(c_int)], c_int),
        "bbx_cg_sample_dev": (
            [hp, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
             c_void_p, c_void_p, c_uint64, c_int, c_double, c_void_p,
             POINTER(c_int), POINTER(c_int)], c_int),
        "bbx_design_matvec_count": (
            [hp, POINTER(c_int64), POINTER(c_int64)], c_int),
        "bbx_design_reset_matvec_count": ([hp], c_int),
        "bbx_design_set_timing": ([hp, c_int], c_int),
        "bbx_design_get_timing": (
            [hp, c_int, POINTER(c_int64), POINTER(c_double)], c_int),
        "bbx_design_reset_timing": ([hp], c_int),
        "bbx_hbm_probe": (
            [c_int, c_int64, c_int, POINTER(c_double), POINTER(c_double)],
            c_int),
        "bbx_chain_create": (
            [hp, c_int, c_void_p, c_void_p, c_int, c_void_p, c_double,
             c_double, c_double, c_double, c_uint64, POINTER(hp)], c_int),
        "bbx_chain_destroy": ([hp], c_int),
        "bbx_chain_set_state": (
            [hp, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
        "bbx_chain_get_state": (
            [hp, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
        "bbx_chain_set_summary": ([hp, c_void_p, c_void_p, c_int64], c_int),
        "bbx_chain_get_summary": (
            [hp, c_void_p, c_void_p, POINTER(c_int64)], c_int),
        "bbx_chain_init_obs_prec": ([hp], c_int),
        "bbx_chain_get_iteration": ([hp, POINTER(c_int64)], c_int),
        "bbx_chain_set_iteration": ([hp, c_int64], c_int),
        "bbx_chain_get_seed": ([hp, POINTER(c_uint64)], c_int),
        "bbx_chain_set_seed": ([hp, c_uint64], c_int),
        "bbx_chain_set_gscale_update": ([hp, c_int], c_int),
        "bbx_chain_eta": ([hp, c_int64, c_void_p, c_void_p], c_int),
        "bbx_chain_get_logp": (
            [hp, POINTER(c_double), POINTER(c_double)], c_int),
        "bbx_chain_run": (
            [hp, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
             c_void_p, c_void_p, c_void_p, c_void_p], c_int),
        "bbx_chain_set_progress": ([hp, c_int, c_void_p, c_void_p], c_int),
        "bbx_chain_run_host": (
            [hp, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
             c_void_p, c_void_p, c_void_p, c_void_p], c_int),
        "bbx_batch_create": ([hp, c_int, c_void_p, POINTER(hp)], c_int),
        "bbx_batch_create_opts": (
            [hp, c_int, c_void_p, ctypes.c_uint, POINTER(hp)], c_int),
        "bbx_batch_predict": ([hp, c_int, POINTER(c_double)], c_int),
        "bbx_batch_unconverged": ([hp, POINTER(c_int)], c_int),
        "bbx_batch_destroy": ([hp], c_int),
        "bbx_batch_run": (
            [hp, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
             c_void_p, c_void_p], c_int),
        "bbx_batch_run_host": (
            [hp, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
             c_void_p, c_void_p], c_int),
        "bbx_batch_dot": ([hp, c_void_p, c_void_p], c_int),
        "bbx_batch_tdot": ([hp, c_void_p, c_void_p], c_int),
        "bbx_batch_bytes": (
            [hp, POINTER(c_int64), POINTER(c_int64)], c_int),
        "bbx_device_polya_gamma": (
            [c_int, c_uint64, c_int64, c_void_p, c_void_p, c_void_p], c_int),
        "bbx_device_tilted_stable": (
            [c_int, c_uint64, c_int64, c_double, c_void_p, c_void_p], c_int),
        "bbx_device_gamma": (
            [c_int, c_uint64, c_int64, c_double, c_void_p], c_int),
        "bbx_device_normal": (
            [c_int, c_uint64, c_uint64, c_int64, c_void_p], c_int),
    }
    for name, (argtypes, restype) in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    return sigs


EXPORTED_SYMBOLS = None


def _one_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.  A process that first
    loads libbbx.so (bound to /opt/rocm's runtime) and later imports torch ends
    up with two HIP runtimes and torch then sees no GPU.  Importing torch first
    makes the dynamic linker hand the already loaded runtime to libbbx.so, so
    both share one.  torch is used for nothing else here
    (BBX_NO_TORCH=1 skips this)."""
    import sys
    if "torch" in sys.modules or os.environ.get("BBX_NO_TORCH") == "1":
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def load():
    """Loads libbbx.so once; raises if it has not been built."""
    global _lib, EXPORTED_SYMBOLS
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BbxError(
                "libbbx.so not found at %s: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C bayes-bridge_amd/csrc`. There is no CPU fallback."
                % LIB_PATH)
        _one_hip_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        lib.bbx_version.restype = c_int
        if lib.bbx_version() != ABI_VERSION:
            # the signatures below are those of include/bbx.h at ABI_VERSION; a
            # stale library would be called with the wrong arity
            raise BbxError(
                "libbbx.so at %s reports version %d, this binding is written "
                "against %d: rebuild it (`make -C bayes-bridge_amd/csrc`)"
                % (LIB_PATH, lib.bbx_version(), ABI_VERSION))
        EXPORTED_SYMBOLS = sorted(_declare(lib))
        _lib = lib
    return _lib


def last_error():
    return load().bbx_last_error().decode("utf-8", "replace")


def check(status):
    """Negative status -> exception; non-negative is returned unchanged."""
    if status < 0:
        raise BbxError("libbbx status %d: %s" % (status, last_error()))
    return status


def device_count():
    n = c_int(0)
    check(load().bbx_device_count(byref(n)))
    return n.value


def builder_threads():
    """Host threads the sparse layout builder uses in this process (affinity
    mask, cgroup CPU quota, LOCAL_WORLD_SIZE; BBX_BUILD_THREADS overrides)."""
    n = c_int(0)
    check(load().bbx_builder_threads(byref(n)))
    return n.value


def hbm_probe(nbytes, reps=20, device=0):
    """(read GB/s, copy GB/s) of a streaming kernel over `nbytes` of HBM."""
    rd, cp = c_double(0.), c_double(0.)
    check(load().bbx_hbm_probe(device, nbytes, reps, byref(rd), byref(cp)))
    return rd.value, cp.value


def require_gpu():
    if device_count() < 1:
        raise BbxError(
            "no HIP device visible: the MI355X path has no CPU fallback")
