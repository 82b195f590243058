"""benchmark/synthetic.py (on-device generator, edges drawn per source) pinned to the summary statistics of the reference's own
generator (benchmark/generate_synthetic_data.py:27-164, edges drawn per destination) at BASELINE config 2's size: fixture
tests/golden/g5_stats_1m.npz written by tests/golden/make_golden_stats.py from the imported reference.  The two draw from the
same generative model with the same destination parameters (same seed, same draws), not the same edges: counts are compared
within sampling bounds, distributions by quantiles."""
import numpy as np
import pytest
import torch

from tests.helpers import load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_generator_reproduces_reference_statistics_at_1m():
    from benchmark.synthetic import generate_matching_problem

    z = load("g5_stats_1m.npz")
    S, D, SP, SEED = z["params"]
    prob = generate_matching_problem(int(S), int(D), float(SP), seed=int(SEED), device=DEV, dtype=torch.float32)
    args = prob["input_args"]
    colptr = args.A.ccol_indices().cpu().numpy().astype(np.int64)
    rows = args.A.row_indices().cpu().numpy()
    a = args.A.values().double().cpu().numpy()
    c = args.c.values().double().cpu().numpy()
    b = args.b_vec.double().cpu().numpy()
    lens = np.diff(colptr)
    nnz_ref = int(z["nnz"])
    # edge counts: Poisson sums -- sd of the total is ~sqrt(1e7) = 3e3; dropping duplicate (source, destination) draws removes
    # ~sum_j p_j^2 / 2 per source (first order in p_j): allow 0.3 %
    assert abs(a.size - nnz_ref) < 3e-3 * nnz_ref, (a.size, nnz_ref)
    assert abs(lens.mean() - float(z["col_len_mean"])) < 0.03
    assert abs(int(lens.max()) - int(z["col_len_max"])) <= 5
    assert 15 <= int((lens == 0).sum()) <= 100 and 15 <= int(z["empty_cols"]) <= 100  # Poisson(1e6 e^-10 = 45)
    h, h_ref = np.bincount(lens, minlength=64)[:64] / lens.size, z["col_len_hist"] / z["col_len_hist"].sum()
    assert 0.5 * np.abs(h - h_ref).sum() < 0.01  # total variation distance of the column-length distributions
    # destinations: same seed -> same breadth p_j, scale s_j, base value v_j, fill ratio rho_j
    row_deg = np.bincount(rows, minlength=int(D))
    q = z["quantiles"]
    inner = (q >= 0.05) & (q <= 0.95)
    assert np.allclose(np.quantile(row_deg, q)[inner], z["row_deg_q"][inner], rtol=0.03, atol=8)  # (a degree of ~110 has a Poisson sd of ~10)
    assert np.allclose(np.quantile(a / -c, q)[inner], z["ratio_q"][inner], rtol=0.02)  # s_j seen through the edges
    # values: c = -min(v u eps, 0.5), a = s c
    assert np.allclose(np.quantile(c, q)[inner], z["c_q"][inner], rtol=0.03) and abs(c.mean() - float(z["c_mean"])) < 0.02 * abs(float(z["c_mean"]))
    assert np.allclose(np.quantile(a, q)[inner], z["a_q"][inner], rtol=0.04)
    assert float(c.min()) == -0.5 and abs((c == -0.5).mean() - float(z["c_at_cap"])) < 3e-4
    # capacities: rho_j (greedy load_j + 1e-8)
    # (a destination's greedy load counts the sources whose LARGEST edge it is: a handful for the lower half of the destinations,
    # so those quantiles carry ~10 % sampling noise; the upper half and the total are tight)
    bq, bq_ref = np.quantile(b, q), z["b_q"]
    assert np.allclose(bq[q >= 0.5][:-1], bq_ref[q >= 0.5][:-1], rtol=0.05) and np.allclose(bq[inner], bq_ref[inner], rtol=0.15)
    assert abs(b.sum() - float(z["b_sum"])) < 0.04 * float(z["b_sum"])  # (dominated by the few largest destinations)
    # structure the kernel relies on: rows strictly increasing inside every column
    k = np.arange(a.size)
    starts = np.zeros(a.size, dtype=bool)
    starts[colptr[:-1][lens > 0]] = True
    assert np.all((np.diff(rows.astype(np.int64)) > 0) | starts[1:])
    del k
