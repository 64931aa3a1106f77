"""CPU: bench.py's byte model of a whole Gibbs iteration (`roofline.iteration`,
`iteration_frac`) against the HBM traffic MEASURED on an MI355X: two rocprofv3
--pmc passes (FETCH_SIZE, WRITE_SIZE) over the last five iterations of a
305-iteration config-3 chain, committed as profiles/r03_iteration_traffic.json
(scripts/iteration_traffic.py, scripts/summarize_iteration_traffic.py)."""
import json
import os

from conftest import ROOT


import pytest


@pytest.mark.parametrize("name", ["r03_iteration_traffic.json",
                                  "r04_iteration_traffic.json"])
def test_iteration_byte_model_is_within_3_percent_of_the_pmc_counters(name):
    import bench
    with open(os.path.join(ROOT, "profiles", name)) as fh:
        prof = json.load(fh)
    assert prof["iters"] == len(prof["n_cg_iter"]) >= 5
    ncg = sum(prof["n_cg_iter"]) / len(prof["n_cg_iter"])
    model = bench.iteration_bytes(
        ncg, prof["dot_bytes"] + prof["tdot_bytes"], prof["dot_bytes"],
        prof["tdot_bytes"], prof["n"], prof["P"])
    # gfx950: FETCH_SIZE counts half the bytes of a 16-byte-per-lane stream
    measured = (2. * prof["fetch_size_kb"] + prof["write_size_kb"]) * 1024. \
        / prof["iters"]
    assert abs(measured - prof["hbm_bytes_per_iteration"]) < 1e-6 * measured
    assert abs(model / measured - 1.) <= .03, (model, measured)
    # the operator's two products are what moves: > 95 % of an iteration
    assert ncg * (prof["dot_bytes"] + prof["tdot_bytes"]) > .9 * measured


def test_dense_cpu_baseline_leg_reports_cores_and_blas():
    """bench.py --config config4's `cpu_baseline`: the oracle chain on NumPy
    dgemv products (dense_matrix.py:42,52) with the BLAS thread count stated;
    run here on a small matrix from a made-up chain state."""
    import numpy as np
    import bench
    rng = np.random.default_rng(0)
    n, p = 3000, 40
    x = rng.standard_normal((n, p)).astype(np.float32)
    offset = x.astype(np.float64).mean(axis=0)
    beta = np.zeros(p)
    beta[:3] = 1.
    y = x.astype(np.float64).dot(beta) + rng.standard_normal(n)
    P = p + 1
    state = (np.zeros(P), 1., np.ones(p), .05, np.zeros(P), np.ones(P), 0)
    out = bench.cpu_baseline_dense(x, offset, y, state, 2, 1)
    assert out["kind"] == "port" and out["value"] > 0
    assert out["cores"] >= 1 and out["unit"] == "Gibbs iters/sec"
    assert out["sample"].startswith("2 Gibbs iterations")
    assert out["blas"] and out["dot_gbs"] > 0 and out["tdot_gbs"] > 0
