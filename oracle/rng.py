"""Random streams of the oracle chain (test infrastructure): the reference's
seeding scheme (random/random.py:17-22) over the oracle's C restatement of the
Polya-Gamma and tilted-stable samplers (oracle/csrc/oracle_samplers.c)."""
import ctypes
import os
import subprocess
from ctypes import c_double, c_int64, c_void_p

import numpy as np
from numpy.random import PCG64

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        lib.oracle_polya_gamma.argtypes = [c_void_p, c_int64, c_void_p,
                                           c_void_p, c_void_p]
        lib.oracle_tilted_stable.argtypes = [c_void_p, c_int64, c_double,
                                             c_void_p, c_void_p]
        for name in ("oracle_csr_matvec", "oracle_csr_rmatvec"):
            getattr(lib, name).argtypes = [c_int64, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p]
        _lib = lib
    return _lib


def _addr(bit_generator):
    return ctypes.cast(bit_generator.ctypes.bit_generator, c_void_p)


def _p(a):
    return a.ctypes.data_as(c_void_p)


class OracleRandom:

    def __init__(self, seed=None):
        self.lib = load()
        self.set_seed(seed)

    def set_seed(self, seed):
        np.random.seed(seed)
        pg_seed = np.random.randint(1, 1 + np.iinfo(np.int32).max)
        ts_seed = np.random.randint(1, 1 + np.iinfo(np.int32).max)
        self.pg = PCG64(pg_seed)
        self.ts = PCG64(ts_seed)

    def polya_gamma(self, shape, tilt):
        shape = np.ascontiguousarray(shape, dtype=np.int32)
        tilt = np.ascontiguousarray(tilt, dtype=np.float64)
        out = np.zeros(tilt.size)
        self.lib.oracle_polya_gamma(_addr(self.pg), tilt.size, _p(shape),
                                    _p(tilt), _p(out))
        return out

    def tilted_stable(self, char_exponent, tilt):
        tilt = np.ascontiguousarray(tilt, dtype=np.float64)
        out = np.zeros(tilt.size)
        self.lib.oracle_tilted_stable(_addr(self.ts), tilt.size,
                                      float(char_exponent), _p(tilt), _p(out))
        return out


class ParallelOracleRandom(OracleRandom):
    """The same samplers on `n_threads` cores, for the multi-core CPU BASELINE
    only (bench.py `cpu_baseline_omp`): the draws of an update are cut into
    equal blocks, every block has its own PCG64 stream (spawned from the
    reference's two seeds), and the C samplers -- which release the GIL under
    ctypes -- run from a thread pool.  Same distributions, not the reference's
    stream (its samplers consume ONE sequential stream each, random.py:17-22)."""

    def __init__(self, seed=None, n_threads=1):
        from concurrent.futures import ThreadPoolExecutor
        self.n_threads = max(1, int(n_threads))
        super().__init__(seed)
        self.pool = ThreadPoolExecutor(self.n_threads)

    def set_seed(self, seed):
        super().set_seed(seed)
        ss = np.random.SeedSequence(seed)
        kids = ss.spawn(2 * self.n_threads)
        self.pg_streams = [PCG64(k) for k in kids[:self.n_threads]]
        self.ts_streams = [PCG64(k) for k in kids[self.n_threads:]]

    def _blocks(self, size):
        step = -(-size // self.n_threads)
        return [(b, min(size, b + step)) for b in range(0, size, step)]

    def polya_gamma(self, shape, tilt):
        shape = np.ascontiguousarray(shape, dtype=np.int32)
        tilt = np.ascontiguousarray(tilt, dtype=np.float64)
        out = np.zeros(tilt.size)

        def work(args):
            k, (b, e) = args
            self.lib.oracle_polya_gamma(_addr(self.pg_streams[k]), e - b,
                                        _p(shape[b:e]), _p(tilt[b:e]),
                                        _p(out[b:e]))
        list(self.pool.map(work, enumerate(self._blocks(tilt.size))))
        return out

    def tilted_stable(self, char_exponent, tilt):
        tilt = np.ascontiguousarray(tilt, dtype=np.float64)
        out = np.zeros(tilt.size)

        def work(args):
            k, (b, e) = args
            self.lib.oracle_tilted_stable(_addr(self.ts_streams[k]), e - b,
                                          float(char_exponent), _p(tilt[b:e]),
                                          _p(out[b:e]))
        list(self.pool.map(work, enumerate(self._blocks(tilt.size))))
        return out


def csr_matvec(X, v):
    """y = X v with the C loop (cross-check of the NumPy/SciPy path)."""
    y = np.zeros(X.shape[0])
    load().oracle_csr_matvec(
        X.shape[0], _p(np.ascontiguousarray(X.indptr, np.int32)),
        _p(np.ascontiguousarray(X.indices, np.int32)),
        _p(np.ascontiguousarray(X.data, np.float64)),
        _p(np.ascontiguousarray(v, np.float64)), _p(y))
    return y


def csr_rmatvec(X, w):
    y = np.zeros(X.shape[1])
    load().oracle_csr_rmatvec(
        X.shape[0], _p(np.ascontiguousarray(X.indptr, np.int32)),
        _p(np.ascontiguousarray(X.indices, np.int32)),
        _p(np.ascontiguousarray(X.data, np.float64)),
        _p(np.ascontiguousarray(w, np.float64)), _p(y))
    return y
