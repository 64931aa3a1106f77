"""CPU: the host-side builder of the LDS-tiled layout and the emulator of the
kernel's walk over it (csrc/tiled_layout.cpp, plain C++ built with g++):
layout == SciPy on the randomised cases the GPU operator test uses, in every
build variant, plus ASan/UBSan and TSan runs of the multi-threaded builder.
The GPU test tests/test_hip_operator.py::test_kernel_equals_cpu_emulator_bitwise
closes the loop: kernel == emulator bit for bit."""
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sparse

from conftest import ROOT
from helpers import TiledLayoutCpu, random_sparse_case


@pytest.fixture(scope="module")
def layout():
    return TiledLayoutCpu()


@pytest.mark.parametrize("case", range(24))
def test_layout_emulation_equals_scipy(layout, case):
    X, binary, rng = random_sparse_case(case)
    X.sort_indices()
    n, p = X.shape
    v, w = rng.standard_normal(p), rng.standard_normal(n)
    Xt = X.T.tocsr()
    Xt.sort_indices()
    ref_v, ref_w = X @ v, Xt @ w
    tol_v = 1e-11 * max(1., np.abs(ref_v).max())
    tol_w = 1e-11 * max(1., np.abs(ref_w).max())
    # packed: value-free ids as groups of five (tiled_layout.hpp packed_slot)
    # forced on / off; left alone the builder takes the form with fewer steps
    variants = [dict(), dict(bank_aware=False), dict(packed=1), dict(packed=0)]
    if case % 4 == 0:
        variants.append(dict(force_PR=128, force_G=2, threads=3))
    # geometries sized for 2 and 4 right-hand sides sharing the pass (batched
    # chains): narrower slices, shorter panels, same sums per right-hand side
    variants += [dict(chains=2), dict(chains=4)]
    quads = {}
    value_free = bool(np.all(X.data == 1.))    # what the builder is handed
    for kw in variants:
        out_v, info_v = layout.matvec(X, v, **kw)
        out_w, info_w = layout.matvec(Xt, w, **kw)
        assert np.abs(out_v - ref_v).max() <= tol_v, (kw, info_v)
        assert np.abs(out_w - ref_w).max() <= tol_w, (kw, info_w)
        if set(kw) <= {'packed'}:
            quads[kw.get('packed', -1)] = (info_v, info_w)
        # groups only for value-free single-chain layouts, and only if asked
        for info in (info_v, info_w):
            assert info['packed'] in (0, 1)
            if not value_free or kw.get('chains', 1) > 1 or kw.get('packed') == 0:
                assert info['packed'] == 0, (kw, info)
            if value_free and kw.get('packed') == 1:
                assert info['packed'] == 1, (kw, info)
        # every layout fits the CU's LDS next to 2 KB of static use
        for info in (info_v, info_w):
            lds = 8 * kw.get('chains', 1) * (info['W'] + 8 + info['PR']
                                             + info['n_extra'])
            assert lds <= 160 * 1024 - 2048 + 8 * 8, info
    # the builder's own choice: groups only where the id stream is large
    # (>= 80 MB of plain ids: tests/test_hip_fullsize.py sees it at 1M x 50k),
    # never for matrices of this size
    for k in range(2):
        auto, plain, grouped = quads[-1][k], quads[0][k], quads[1][k]
        assert not auto['packed'] and auto['n_quad'] == plain['n_quad']
        if value_free:
            assert grouped['n_quad'] <= plain['n_quad'] + plain['n_slice']
    # binary designs: the emulator adds whole numbers exactly
    if binary and np.all(v == np.round(v)):
        assert np.array_equal(out_v, ref_v)


def test_integer_vectors_give_exact_sums(layout):
    """With integer-valued inputs every partial sum is exact, so the layout
    (row splitting, fold order, slab order) must reproduce SciPy bit for bit."""
    from bayesbridge_amd import simulate
    X = simulate.simulate_binary_csr_fast(9000, 20000, .004, seed=7)
    rng = np.random.default_rng(1)
    v = rng.integers(-50, 50, X.shape[1]).astype(np.float64)
    w = rng.integers(-50, 50, X.shape[0]).astype(np.float64)
    Xt = X.T.tocsr()
    Xt.sort_indices()
    for kw in (dict(), dict(force_PR=256, force_G=2), dict(chains=2),
               dict(chains=4), dict(packed=1), dict(packed=0),
               dict(packed=1, force_PR=256, force_G=2)):
        assert np.array_equal(layout.matvec(X, v, **kw)[0], X @ v)
        assert np.array_equal(layout.matvec(Xt, w, **kw)[0], Xt @ w)


def packed_edge_case_matrix():
    """Rows built to hit every rule of the packed groups (csrc/tiled_layout.hpp
    packed_slot): columns on both sides of the zero slots (4094 | 4095, 8189 |
    8190, ...), gaps of exactly 4095 and 4096 slots, duplicate column entries
    (a zero delta between REAL entries: counted twice, as in SciPy), rows of 1
    to 11 entries (groups that end on a zero slot after 1 ... 4 entries), the
    last column of a block and of the matrix."""
    C = 13000
    rows = [
        [0], [4094], [4095], [4096], [C - 1],
        [4094, 4095, 4096], [8189, 8190, 8191, 8192],
        [0, 4095], [0, 4096], [1, 4096, 8191],             # gaps 4095 / 4096 / ...
        [5, 5], [7, 7, 7, 9, 9, 4095, 4095],               # duplicates
        list(range(100, 111)), list(range(4090, 4101)),    # 11 entries, across a zero slot
        [3, 4000, 8000, 12000, C - 1], [2, 3, 4, 5, 6, 7], [12283, 12284, 12285, 12286],
        [],
    ]
    rows = rows * 9                                         # several slices
    # background: every column holds an entry (the drop-in package removes
    # constant columns like the reference does, all-zero ones included)
    rows += [list(range(r, C, 1625)) for r in range(1625)]
    indptr = np.concatenate(([0], np.cumsum([len(r) for r in rows]))).astype(np.int32)
    indices = np.array([c for r in rows for c in r], dtype=np.int32)
    return sparse.csr_matrix((np.ones(len(indices)), indices, indptr),
                             shape=(len(rows), C))


def test_packed_groups_edge_cases(layout):
    A = packed_edge_case_matrix()
    assert not A.has_canonical_format                # the duplicates are in
    rng = np.random.default_rng(8)
    v = rng.integers(-40, 40, A.shape[1]).astype(np.float64)
    ref = A @ v                                      # csr_matvec adds duplicates
    for kw in (dict(packed=1), dict(packed=1, bank_aware=False),
               dict(packed=1, force_PR=64), dict(packed=0)):
        out, info = layout.matvec(A, v, **kw)
        assert info['packed'] == kw['packed'] and info['W'] > 12285
        assert np.array_equal(out, ref), kw
    # the transposed orientation
    At = sparse.csr_matrix(A.T)
    At.sort_indices()
    w = rng.integers(-40, 40, A.shape[0]).astype(np.float64)
    out, info = layout.matvec(At, w, packed=1)
    assert info['packed'] == 1 and np.array_equal(out, At @ w)


def test_bank_aware_order_lowers_lds_conflicts(layout):
    """The builder's entry order inside rows: fewer LDS cycles per gather
    (1.0 = conflict free; ascending ids give ~3.4) and the same sums."""
    from bayesbridge_amd import simulate
    # big panels (4096 rows: two slices per wave, no chunking for occupancy),
    # ~60 entries per row segment, like the tiles of the headline config
    X = simulate.simulate_binary_csr_fast(16384, 16000, .004, seed=3)
    Xt = X.T.tocsr()
    Xt.sort_indices()
    rng = np.random.default_rng(2)
    for A in (X, Xt):
        x = rng.standard_normal(A.shape[1])
        plain, ip = layout.matvec(A, x, bank_aware=False, force_PR=4096,
                                  packed=0)
        tuned, it = layout.matvec(A, x, bank_aware=True, force_PR=4096,
                                  packed=0)
        assert np.abs(plain - tuned).max() <= 1e-12 * np.abs(plain).max()
        assert ip['n_quad'] == it['n_quad']          # not a byte more
        assert ip['gather_cycles'] > 2.5
        assert it['gather_cycles'] < .7 * ip['gather_cycles'], (ip, it)


def test_builder_under_address_undefined_and_thread_sanitizers():
    """make sanitize: the self-test driver (random matrices, 4 builder threads,
    value-free/valued layouts, emulation vs a plain CSR product) under
    -fsanitize=address,undefined and -fsanitize=thread."""
    res = subprocess.run(
        ['make', '-C', os.path.join(ROOT, 'bayes-bridge_amd', 'csrc'),
         'sanitize'], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert 'FAILED' not in res.stdout + res.stderr
    assert 'WARNING: ThreadSanitizer' not in res.stderr
    assert 'ERROR: AddressSanitizer' not in res.stderr
    assert 'runtime error' not in res.stderr
    assert res.stdout.count('max err') >= 14        # both binaries ran


def test_builder_threads_follow_quota_and_local_ranks():
    """The layout builder's worker count: affinity mask capped by the cgroup
    CPU quota, shared among the LOCAL_WORLD_SIZE ranks of a node, capped at 64;
    BBX_BUILD_THREADS overrides.  (Eight ranks with 64 threads each under the
    GPU boxes' 16-CPU quota is the regime that collapsed in
    profiles/r03_omp_threads.txt.)"""
    import sys
    from oracle.omp_baseline import usable_cores
    lib = os.path.join(ROOT, 'bayes-bridge_amd', 'libbbx_layout.so')
    code = ("import ctypes,sys; l=ctypes.CDLL(%r); "
            "print(l.bbx_layout_builder_threads(int(sys.argv[1])))" % lib)

    def ask(max_threads=0, **env):
        e = {k: v for k, v in os.environ.items()
             if k not in ('LOCAL_WORLD_SIZE', 'BBX_BUILD_THREADS')}
        e.update({k: str(v) for k, v in env.items()})
        return int(subprocess.check_output(
            [sys.executable, '-c', code, str(max_threads)], env=e, text=True))

    cores = usable_cores()          # same rule, restated in Python
    assert ask() == min(64, cores)
    assert ask(LOCAL_WORLD_SIZE=8) == max(1, min(64, cores // 8))
    assert ask(LOCAL_WORLD_SIZE=10 ** 6) == 1
    assert ask(max_threads=3) == min(3, cores)
    assert ask(BBX_BUILD_THREADS=5, LOCAL_WORLD_SIZE=8) == 5
    assert ask(BBX_BUILD_THREADS=500) == 64


def test_cost_model_orders_the_batch_widths_like_the_measurements():
    """bbx_batch_predict (csrc/spmv_tiled.hip tiled_batch_predict) prices a batch
    of K chains as K x the single-chain layouts' model cost over the K-layouts'
    (csrc/tiled_layout.cpp shape_cost); bbx_batch_create refuses a width priced
    below 1.  Without a GPU: the same arithmetic on row-pointer arrays with the
    headline design's statistics (1M x 50k, column frequencies
    0.5 Beta(.5, .5 (.5 / f - 1)), ~100 entries per row).  Measured on the
    MI355X (profiles/r04_bench.json): pairs 1.38x, fours 0.98x at this size;
    fours 1.54x at 100k x 10k."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, 'bayes-bridge_amd',
                                   'libbbx_layout.so'))
    lib.bbx_layout_model_cost.restype = ctypes.c_double
    lib.bbx_layout_model_cost.argtypes = [ctypes.c_int64] * 3 + [
        ctypes.c_void_p, ctypes.c_int]

    def predicted(n, p, f, K, seed=0):
        rng = np.random.default_rng(seed)
        freq = .5 * rng.beta(.5, .5 * (.5 / f - 1), p)
        col_n = np.ceil(n * freq).astype(np.int64)
        nnz = int(col_n.sum())
        t_ptr = np.concatenate([[0], np.cumsum(col_n)]).astype(np.int32)
        row_n = rng.poisson(nnz / n, n)
        row_n[-1] += nnz - row_n.sum()
        x_ptr = np.concatenate([[0], np.cumsum(row_n)]).astype(np.int32)
        cost = {}
        for k in (1, K):
            cost[k] = lib.bbx_layout_model_cost(n, p, nnz, x_ptr.ctypes.data, k) \
                + lib.bbx_layout_model_cost(p, n, nnz, t_ptr.ctypes.data, k)
            assert cost[k] > 0
        return K * cost[1] / cost[K]
    big2, big4 = predicted(1000000, 50000, .002, 2), predicted(1000000, 50000, .002, 4)
    assert big2 > 1. > big4, (big2, big4)          # pairs accepted, fours refused
    small2, small4 = predicted(100000, 10000, .01, 2), predicted(100000, 10000, .01, 4)
    assert small4 > small2 > 1., (small2, small4)  # both pay on the small design
    assert lib.bbx_layout_model_cost(10, 10, 10, None, 3) < 0   # K = 3: no layout


def _transpose64(lib, A, vals=True, threads=4, indptr=None, indices=None):
    import ctypes
    R, C = A.shape
    indptr = np.ascontiguousarray(A.indptr if indptr is None else indptr,
                                  dtype=np.int64)
    indices = np.ascontiguousarray(A.indices if indices is None else indices,
                                   dtype=np.int64)
    data = np.ascontiguousarray(A.data, dtype=np.float64)
    nnz = len(data)
    t_ptr = np.empty(C + 1, dtype=np.int64)
    t_idx = np.empty(max(nnz, 1), dtype=np.int32)
    t_val = np.empty(max(nnz, 1), dtype=np.float64)
    lib.bbx_layout_transpose64.argtypes = (
        [ctypes.c_int64] * 2 + [ctypes.c_void_p] * 3 + [ctypes.c_int]
        + [ctypes.c_void_p] * 3)
    st = lib.bbx_layout_transpose64(
        R, C, indptr.ctypes.data, indices.ctypes.data,
        data.ctypes.data if vals else None, threads, t_ptr.ctypes.data,
        t_idx.ctypes.data, t_val.ctypes.data if vals else None)
    return st, t_ptr, t_idx[:nnz], t_val[:nnz]


@pytest.mark.parametrize("shape,density,threads", [
    ((1, 1), 1., 1), ((7, 3), .5, 4), ((300, 1000), .02, 3),
    ((2000, 40), .3, 8), ((513, 129), .0, 2)])
def test_host_transposition_of_64bit_csr_equals_scipys(shape, density, threads):
    """bbx_design_create_csr64's host path (2^31 or more stored entries) builds
    the CSR of X^T with a threaded stable counting sort (csrc/tiled_layout.cpp
    transpose_csr_host): row pointers, row ids in ascending order and values
    must be those of scipy's X.T.tocsr() -- what the reference's Tdot multiplies
    with (design_matrix/sparse_matrix.py:103-129)."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, 'bayes-bridge_amd',
                                   'libbbx_layout.so'))
    rng = np.random.default_rng(5)
    A = sparse.random(shape[0], shape[1], density=density, format='csr',
                      random_state=rng, dtype=np.float64)
    A.sort_indices()
    st, t_ptr, t_idx, t_val = _transpose64(lib, A, threads=threads)
    assert st == 0
    T = A.T.tocsr()
    T.sort_indices()
    assert np.array_equal(t_ptr, T.indptr)
    assert np.array_equal(t_idx, T.indices)
    assert np.array_equal(t_val, T.data)
    st, t_ptr, t_idx, _ = _transpose64(lib, A, vals=False, threads=threads)
    assert st == 0 and np.array_equal(t_ptr, T.indptr)
    assert np.array_equal(t_idx, T.indices)


def test_host_structure_check_of_64bit_csr_reports_what_the_device_check_does():
    """Bits of check_csr64_host = those of validate_csr_kernel (csrc/api.hip):
    1 row pointers, 2 column id out of range, 4 columns of a row not ascending;
    duplicates are allowed (SciPy's csr_matvec adds them up)."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, 'bayes-bridge_amd',
                                   'libbbx_layout.so'))
    A = sparse.csr_matrix(np.array([[1., 0, 2, 0], [0, 3, 0, 4], [5, 0, 0, 6]]))

    def status(indptr=None, indices=None):
        return _transpose64(lib, A, indptr=indptr, indices=indices)[0]
    assert status() == 0
    assert status(indptr=[1, 2, 4, 6]) & 1
    assert status(indptr=[0, 4, 2, 6]) & 1
    assert status(indices=[0, 2, 1, 4, 0, 3]) & 2
    assert status(indices=[0, 2, 1, -1, 0, 3]) & 2
    assert status(indices=[2, 0, 1, 3, 0, 3]) & 4
    assert status(indices=[0, 0, 1, 3, 3, 3]) == 0     # duplicates
