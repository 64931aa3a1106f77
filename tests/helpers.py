"""Shared seeded inputs for the parity tests (no reference, no GPU needed)."""
import numpy as np
import scipy.sparse as sparse

from bayesbridge_amd import simulate


def mixed_design(n, p, binary_frac=.5, freq=.1, seed=0):
    """Mixed Gaussian/binary CSR design with the reference's distribution
    (tests/helper.py:8-16 uses simulate_design(n, p, binary_frac=.9))."""
    return simulate.simulate_design_csr(n, p, binary_frac=binary_frac,
                                        binary_pred_freq=freq, seed=seed)


def cg_inputs(n, P, n_unshrunk=1, seed=0, flat_intercept=True,
              lam_log_sd=1.5):
    """Plausible inputs of one CG draw: Omega ~ PG-like positives, prior
    precisions spanning several decades (bridge prior), a warm start."""
    rng = np.random.default_rng(seed)
    omega = rng.gamma(2., .15, n) + 1e-3
    lam = np.exp(rng.normal(0., lam_log_sd, P))
    sd_prior = .05 * lam / np.sqrt(1 + (.05 * lam / 2.) ** 2)
    phi = 1. / sd_prior
    if flat_intercept and n_unshrunk > 0:
        phi[0] = 0.
    z = rng.normal(0., 3., P)
    x0 = rng.normal(0., .05, P)
    sd = .5 + rng.random(P)
    eta1 = rng.standard_normal(n)
    eta2 = rng.standard_normal(P)
    return dict(obs_prec=omega, prior_prec_sqrt=phi, z=z, coef_cg_init=x0,
                coef_scaled_sd=sd, randn_n=eta1, randn_P=eta2,
                n_unshrunk=n_unshrunk)


def config2_small_problem(golden_dir,
                          name='chain_logit_binary_20000x1000_summary.npz'):
    """The BASELINE config 2 problem of a fixture under tests/golden/ (scaled
    down: chain_logit_binary_20000x1000_summary.npz; full size:
    chain_logit_binary_100000x10000_first10.npz), regenerated without the
    reference (bayesbridge_amd.simulate replays the reference's RNG calls) and
    checked against the checksums stored in the fixture."""
    import os
    from bayesbridge_amd import simulate
    g = np.load(os.path.join(golden_dir, name))
    n, p = (int(v) for v in g['shape'])
    X = simulate.simulate_design_csr(n, p, binary_frac=1.,
                                     binary_pred_freq=float(g['freq']),
                                     seed=111)
    assert X.nnz == int(g['nnz'])
    assert np.array_equal(X.indptr[-4:], g['indptr_tail'])
    chk = (X.indices.astype(np.int64) * (np.arange(X.nnz) % 1009 + 1)).sum()
    assert chk == int(g['indices_checksum'])
    beta = simulate.demo_beta(p)
    n_success, n_trial = simulate.simulate_outcome(X, beta, 'logit', seed=1)
    assert n_success.sum() == g['n_success_sum']
    assert np.array_equal(n_success[:32], g['n_success_head'])
    assert np.array_equal(n_trial[:32], g['n_trial_head'])
    return g, X, (n_success, n_trial)


def random_sparse_case(case):
    """Randomised shapes and patterns (shared by the CPU layout tests and the
    GPU operator tests): one to several column blocks (p around the
    16128-column slice width), panels from a handful of rows to several
    thousand, rows from empty to dense, binary and valued entries, duplicates.
    Returns (X csr, binary flag, rng)."""
    rng = np.random.default_rng(1000 + case)
    n = int(rng.choice([3, 17, 130, 1000, 4097, 9000]))
    p = int(rng.choice([2, 65, 900, 16128, 16130, 33000, 50000]))
    density = float(rng.choice([.0005, .004, .03])) if p > 1000 \
        else float(rng.choice([.02, .2, .7]))
    nnz = max(1, int(n * p * density))
    rows = rng.integers(0, n, nnz)
    # skewed columns: a few hot ones, many rare ones
    cols = np.minimum((p * rng.random(nnz) ** 3).astype(np.int64), p - 1)
    binary = bool(case % 2)
    vals = np.ones(nnz) if binary else rng.standard_normal(nnz)
    X = sparse.coo_matrix((vals, (rows, cols)), shape=(n, p)).tocsr()
    if not binary or case % 3 == 0:
        X.sum_duplicates()
    else:
        # keep duplicates as separate stored entries (legal CSR, values add up)
        order = np.lexsort((cols, rows))
        indptr = np.zeros(n + 1, dtype=np.int32)
        np.add.at(indptr, rows + 1, 1)
        X = sparse.csr_matrix((vals[order], cols[order].astype(np.int32),
                               np.cumsum(indptr).astype(np.int32)),
                              shape=(n, p))
    return X, binary, rng


class TiledLayoutCpu:
    """ctypes front of libbbx_layout.so: the host-side builder of the tiled
    layout and the CPU emulator of tiled_spmv_kernel's walk (csrc/
    tiled_layout.cpp built with g++, no GPU needed)."""

    def __init__(self):
        import ctypes
        import os
        import subprocess
        from ctypes import POINTER, c_double, c_int, c_int64, c_void_p
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(root, 'bayes-bridge_amd', 'libbbx_layout.so')
        if not os.path.exists(path):
            subprocess.check_call(
                ['make', '-C', os.path.join(root, 'bayes-bridge_amd', 'csrc'),
                 '../libbbx_layout.so'], stdout=subprocess.DEVNULL)
        self.lib = ctypes.CDLL(path)
        self.lib.bbx_layout_emulate.argtypes = (
            [c_int64] * 3 + [c_void_p] * 3 + [c_int] * 7 + [c_void_p] * 2
            + [POINTER(c_int64), POINTER(c_double)])

    def matvec(self, A, x, bank_aware=True, force_PR=0,
               force_G=0, threads=4, chains=1, force_blocks=0, packed=-1):
        """(A x, info dict) through the layout + emulator; A is R x C CSR
        with ascending column indices inside each row.  packed: -1 the
        builder's choice between plain ids and groups of five, 0 / 1 forced."""
        import ctypes
        A = sparse.csr_matrix(A)
        R, C = A.shape
        indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
        indices = np.ascontiguousarray(A.indices, dtype=np.int32)
        if indices.size == 0:
            indices = np.zeros(1, dtype=np.int32)
        binary = bool(np.all(A.data == 1.))
        data = None if binary else np.ascontiguousarray(A.data, np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty(R)
        info = (ctypes.c_int64 * 9)()
        cyc = ctypes.c_double()
        st = self.lib.bbx_layout_emulate(
            R, C, A.nnz, indptr.ctypes.data, indices.ctypes.data,
            None if data is None else data.ctypes.data,
            int(bank_aware), int(force_PR), int(force_G), int(force_blocks),
            int(threads), int(chains), int(packed),
            x.ctypes.data, out.ctypes.data, info, ctypes.byref(cyc))
        if st != 0:
            raise RuntimeError("tiled layout could not be built")
        keys = ('W', 'n_block', 'PR', 'G', 'n_quad', 'n_slice', 'n_extra',
                'split_T', 'packed')
        meta = dict(zip(keys, (int(v) for v in info)))
        meta['gather_cycles'] = cyc.value
        return out, meta
