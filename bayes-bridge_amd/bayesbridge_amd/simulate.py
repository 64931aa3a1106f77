"""Synthetic designs and outcomes with the distributions of the reference's
`simulate_data.py`, generated straight into CSR.

Why this exists: `simulate_data.simulate_binary_design` materialises a dense
`np.zeros((n_obs, n_binary_pred))` (simulate_data.py:113), i.e. 400 GB for the
headline 1M x 50k benchmark.  `simulate_design_csr` issues the SAME calls to the
global NumPy stream in the SAME order (`randn` for the dense block,
`beta(a, b, n_binary)`, then one `choice(n_obs, nnz_j, replace=False)` per
column; simulate_data.py:65-70,109-116) but appends row indices instead of
writing a dense array, so for a given seed it returns exactly the matrix
`simulate_design(..., format_='sparse')` would.  `simulate_binary_csr_fast`
draws the same distribution (column frequency 0.5*Beta(.5, .5(.5/f-1)),
nnz_j = ceil(n f_j) distinct uniform rows) with vectorised NumPy `Generator`
calls -- not stream-identical, used where the exact replay is too slow.
"""
import math

import numpy as np
import scipy.sparse as sparse


def binary_column_counts(n_obs, freq):
    """nnz_j = ceil(n * freq_j) (simulate_data.py:115)."""
    return np.array([math.ceil(n_obs * f) for f in freq], dtype=np.int64)


def simulate_design_csr(n_obs, n_pred, binary_frac=0., binary_pred_freq=.1,
                        max_freq_per_col=.5, seed=None):
    """Exact replay of simulate_design(n_obs, n_pred, binary_frac,
    binary_pred_freq=..., format_='sparse', seed=seed) without categorical
    predictors (simulate_data.py:29-63)."""
    if seed is not None:
        np.random.seed(seed)
    n_dense = int(n_pred * (1 - binary_frac))
    n_binary = n_pred - n_dense
    X_dense = np.random.randn(n_obs, n_dense)
    if n_binary == 0:
        return sparse.csr_matrix(X_dense)
    a = .5
    b = a * (max_freq_per_col / binary_pred_freq - 1)
    freq = max_freq_per_col * np.random.beta(a, b, n_binary)
    counts = binary_column_counts(n_obs, freq)
    rows = np.empty(int(counts.sum()), dtype=np.int32)
    colptr = np.concatenate(([0], np.cumsum(counts)))
    for j in range(n_binary):
        rows[colptr[j]:colptr[j + 1]] = np.random.choice(
            n_obs, int(counts[j]), replace=False)
    X_bin = sparse.csc_matrix(
        (np.ones(len(rows)), rows, colptr), shape=(n_obs, n_binary))
    X = sparse.hstack((sparse.csr_matrix(X_dense), X_bin)).tocsr()
    X.sort_indices()
    return X


def simulate_binary_csr_fast(n_obs, n_pred, binary_pred_freq=.1,
                             max_freq_per_col=.5, seed=0):
    """Same distribution as the binary block of simulate_design, vectorised
    (NumPy Generator, PCG64): rows are drawn with replacement, duplicates are
    dropped and topped up until every column has exactly nnz_j distinct rows."""
    rng = np.random.default_rng(seed)
    a = .5
    b = a * (max_freq_per_col / binary_pred_freq - 1)
    freq = max_freq_per_col * rng.beta(a, b, n_pred)
    counts = np.ceil(n_obs * freq).astype(np.int64)
    counts = np.maximum(counts, 0)
    need = counts.copy()
    keys = np.empty(0, dtype=np.int64)
    while need.sum() > 0:
        cols = np.repeat(np.arange(n_pred, dtype=np.int64), need)
        rows = rng.integers(0, n_obs, size=len(cols), dtype=np.int64)
        keys = np.unique(np.concatenate((keys, cols * n_obs + rows)))
        have = np.bincount(keys // n_obs, minlength=n_pred)
        need = counts - have
    cols = (keys // n_obs).astype(np.int32)
    rows = (keys % n_obs).astype(np.int32)
    X = sparse.csc_matrix(
        (np.ones(len(rows)), rows, np.concatenate(([0], np.cumsum(counts)))),
        shape=(n_obs, n_pred)).tocsr()
    X.sort_indices()
    return X


def demo_beta(n_pred):
    """True coefficients of the reference demo (demo.ipynb cell 5)."""
    beta = np.zeros(n_pred)
    beta[:5] = 1.5
    beta[5:10] = 1.
    beta[10:15] = .5
    return beta


def simulate_outcome(X, beta, model, intercept=0., n_trial=None, seed=None):
    """simulate_data.py:8-27 for the linear and logit families."""
    if seed is not None:
        np.random.seed(seed)
    if model == 'linear':
        return intercept + X.dot(beta) + np.random.randn(X.shape[0])
    if model == 'logit':
        if n_trial is None:
            n_trial = np.ones(X.shape[0])
        prob = 1 / (1 + np.exp(- intercept - X.dot(beta)))
        n_success = np.random.binomial(n_trial.astype(np.int32), prob)
        return n_success, n_trial
    raise NotImplementedError(model)


def simulate_binary_csr_device(n_obs, n_pred, binary_pred_freq=.1,
                               max_freq_per_col=.5, seed=0, device='cuda:0'):
    """The binary design of simulate_design (simulate_data.py:100-117)
    generated directly in HBM with torch: column frequencies
    0.5*Beta(.5, .5(.5/f - 1)) (host, NumPy Generator(PCG64(seed))), then for
    every column exactly nnz_j = ceil(n f_j) DISTINCT uniformly drawn rows
    (device, torch Philox generator seeded with `seed`): draw with replacement,
    drop duplicates, top up until exact.  Same distribution as the reference's
    per-column `choice(n, nnz_j, replace=False)`; not stream-identical.

    Returns (indptr int32 [n+1], indices int32 [nnz]) as torch tensors on
    `device`, column indices ascending inside each row; all values are 1.0.
    """
    import torch
    rng = np.random.default_rng(seed)
    a = .5
    b = a * (max_freq_per_col / binary_pred_freq - 1)
    freq = max_freq_per_col * rng.beta(a, b, n_pred)
    counts_np = np.ceil(n_obs * freq).astype(np.int64)
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    counts = torch.from_numpy(counts_np).to(dev)
    need = counts.clone()
    col_ids = torch.arange(n_pred, device=dev, dtype=torch.int64)
    keys = torch.empty(0, dtype=torch.int64, device=dev)
    while int(need.sum().item()) > 0:
        cols = torch.repeat_interleave(col_ids, need)
        rows = torch.randint(0, n_obs, (cols.numel(),), generator=gen,
                             device=dev, dtype=torch.int64)
        keys = torch.unique(torch.cat((keys, cols * n_obs + rows)))
        del cols, rows
        have = torch.bincount(torch.div(keys, n_obs, rounding_mode='floor'),
                              minlength=n_pred)
        need = counts - have
    cols = torch.div(keys, n_obs, rounding_mode='floor')
    rows = keys - cols * n_obs
    del keys
    key2, _ = torch.sort(rows * n_pred + cols)
    row_counts = torch.bincount(rows, minlength=n_obs)
    del rows, cols
    rows2 = torch.div(key2, n_pred, rounding_mode='floor')
    indices = (key2 - rows2 * n_pred).to(torch.int32)
    del key2, rows2
    indptr = torch.zeros(n_obs + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(row_counts, 0)
    return indptr.to(torch.int32), indices
