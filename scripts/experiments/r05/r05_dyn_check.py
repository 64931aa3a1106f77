"""Round 5: dynamic dispatch inside a workgroup (BBX_TILED_DYN=1) against the
static per-wave schedules on the same designs: results must be equal bit for
bit (the slices are the same, only who sums them differs), and equal to the
CPU emulator's.  Usage: python scripts/r05_dyn_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("BBX_PACKAGE_DIR",
                                  os.path.join(ROOT, "bayes-bridge_amd")))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from bayesbridge_amd import HipSparseDesignMatrix, simulate
from helpers import TiledLayoutCpu, random_sparse_case


def build(A, dyn, pack=None, **kw):
    os.environ['BBX_TILED_DYN'] = str(dyn)
    if pack is not None:
        os.environ['BBX_TILED_PACK'] = str(pack)
    try:
        return HipSparseDesignMatrix(A.copy(), storage='tiled', **kw)
    finally:
        os.environ.pop('BBX_TILED_DYN', None)
        os.environ.pop('BBX_TILED_PACK', None)


layout = TiledLayoutCpu()
n_checked = 0
for case in list(range(1, 24, 2)) + ['c2small', 'mid']:
    if case == 'c2small':
        A = simulate.simulate_binary_csr_fast(20000, 3000, .02, seed=11)
        rng = np.random.default_rng(3)
    elif case == 'mid':
        A = simulate.simulate_binary_csr_fast(60000, 40000, .003, seed=12)
        rng = np.random.default_rng(4)
    else:
        A, binary, rng = random_sparse_case(case)
        A.sort_indices()
        if not np.all(A.data == 1.):
            continue
    n, p = A.shape
    if p < 8 or np.any(np.diff(A.tocsc().indptr) == 0) or \
            np.any(np.diff(A.tocsc().indptr) == n):
        continue          # (the wrapper drops constant columns: other shape)
    At = A.T.tocsr()
    At.sort_indices()
    v, w = rng.standard_normal(p), rng.standard_normal(n)
    for pack in (0, 1):
        hs = build(A, 0, pack, center_predictor=False, add_intercept=False)
        hd = build(A, 1, pack, center_predictor=False, add_intercept=False)
        emu_v, info_x = layout.matvec(A, v, packed=pack, dynamic=1)
        emu_w, info_t = layout.matvec(At, w, packed=pack, dynamic=1)
        for _ in range(3):
            dv, dw = hd.dot(v), hd.Tdot(w)
            assert np.array_equal(dv, hs.dot(v)), (case, pack)
            assert np.array_equal(dw, hs.Tdot(w)), (case, pack)
            assert np.array_equal(dv, emu_v), (case, pack)
            assert np.array_equal(dw, emu_w), (case, pack)
        # centred + intercept: the 8-byte slice fill (v + 1 is not 16-byte aligned)
        hc = build(A, 1, pack, center_predictor=True, add_intercept=True)
        hcs = build(A, 0, pack, center_predictor=True, add_intercept=True)
        v1 = rng.standard_normal(hc.shape[1])
        assert np.array_equal(hc.dot(v1), hcs.dot(v1)), (case, pack)
        assert np.array_equal(hc.Tdot(w), hcs.Tdot(w)), (case, pack)
        n_checked += 1
    print("case", case, A.shape, A.nnz, "dyn == static == emulator",
          info_x['dyn'], info_t['dyn'], flush=True)
print("OK:", n_checked, "layouts")
