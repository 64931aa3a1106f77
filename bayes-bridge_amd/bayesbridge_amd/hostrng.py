"""Host-side random streams of the exact-seed parity mode.

Same seeding and the same three streams as the reference's `BasicRandom`
(random/random.py:5-41): the global NumPy MT19937 stream plus two PCG64 bit
generators for the Polya-Gamma and tilted-stable samplers, which here live in
libbbx_hostrng.so (csrc/hostrng.cpp, csrc/samplers.hpp) instead of Cython."""
import ctypes
import os
from ctypes import c_int64, c_void_p

import numpy as np
from numpy.random import PCG64

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get(
    "BBX_HOSTRNG_LIBRARY",
    os.path.join(os.path.dirname(_HERE), "libbbx_hostrng.so"))
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libbbx_hostrng.so not found at %s (run __graft_entry__.build())"
                % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name in ("bbx_host_polya_gamma", "bbx_host_tilted_stable"):
            fn = getattr(lib, name)
            fn.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]
            fn.restype = ctypes.c_int
        _lib = lib
    return _lib


def _bitgen_address(bit_generator):
    return ctypes.cast(bit_generator.ctypes.bit_generator, c_void_p)


class ReferenceRandom():

    def __init__(self, seed=None):
        self.np_random = np.random
        self._lib = load()
        self.set_seed(seed)

    def set_seed(self, seed):
        self.np_random.seed(seed)                           # random.py:17-22
        pg_seed = np.random.randint(1, 1 + np.iinfo(np.int32).max)
        ts_seed = np.random.randint(1, 1 + np.iinfo(np.int32).max)
        self.pg_bitgen = PCG64(pg_seed)
        self.ts_bitgen = PCG64(ts_seed)

    def get_state(self):
        return {'numpy': self.np_random.get_state(),
                'tilted_stable': self.ts_bitgen.state,
                'polya_gamma': self.pg_bitgen.state}

    def set_state(self, state):
        self.np_random.set_state(state['numpy'])
        self.ts_bitgen.state = state['tilted_stable']
        self.pg_bitgen.state = state['polya_gamma']

    def polya_gamma(self, shape, tilt):
        """rand_polyagamma (polya_gamma.pyx:40-74)."""
        if not np.issubdtype(np.asarray(shape).dtype, np.integer):
            raise ValueError('Shape parameter must be integers.')
        shape = np.ascontiguousarray(shape, dtype=np.int32)
        tilt = np.ascontiguousarray(tilt, dtype=np.float64)
        if shape.size != tilt.size:
            raise ValueError('Input arrays must be of the same length.')
        out = np.zeros(tilt.size)
        st = self._lib.bbx_host_polya_gamma(
            _bitgen_address(self.pg_bitgen), tilt.size,
            shape.ctypes.data_as(c_void_p), tilt.ctypes.data_as(c_void_p),
            out.ctypes.data_as(c_void_p))
        if st != 0:
            raise ValueError("invalid Polya-Gamma parameters")
        return out

    def tilted_stable(self, char_exponent, tilt):
        """ExpTiltedStableDist.sample (tilted_stable.pyx:65-134)."""
        tilt = np.ascontiguousarray(tilt, dtype=np.float64)
        a = np.ascontiguousarray(
            np.broadcast_to(np.asarray(char_exponent, dtype=np.float64),
                            tilt.shape))
        if not np.all(a < 1):
            raise ValueError('Characteristic exponent must be smaller than 1.')
        if not np.all(tilt > 0):
            raise ValueError('Tilting parameter must be positive.')
        out = np.zeros(tilt.size)
        st = self._lib.bbx_host_tilted_stable(
            _bitgen_address(self.ts_bitgen), tilt.size,
            a.ctypes.data_as(c_void_p), tilt.ctypes.data_as(c_void_p),
            out.ctypes.data_as(c_void_p))
        if st != 0:
            raise ValueError("invalid tilted-stable parameters")
        return out
