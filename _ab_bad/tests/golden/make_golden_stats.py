#!/usr/bin/env python3
"""Fixture G5-stats (runs ONLY in the build container): summary statistics of the reference's own synthetic generator
(benchmark/generate_synthetic_data.py:27-164) at the size of BASELINE config 2 -- 1M sources x 10k destinations, sparsity 1e-3,
seed 42, float32 -- written to tests/golden/g5_stats_1m.npz.  benchmark/synthetic.py (the on-device generator with the same
generative model, drawn per source instead of per destination) is pinned to them in tests/test_gpu_generator.py.
Re-run with:  python tests/golden/make_golden_stats.py   (about 15 s)"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
stub = tempfile.mkdtemp(prefix="mlflow_stub_")
os.makedirs(os.path.join(stub, "mlflow"), exist_ok=True)
open(os.path.join(stub, "mlflow", "__init__.py"), "w").close()
sys.path[:0] = [stub, os.path.join(REF, "src"), os.path.join(REF, "benchmark")]

import torch  # noqa: E402
from generate_synthetic_data import generate_synthetic_matching_input_args  # noqa: E402

S, D, SP, SEED = 1_000_000, 10_000, 1e-3, 42
args = generate_synthetic_matching_input_args(S, D, SP, device="cpu", dtype=torch.float32, seed=SEED)
A, C, b = args.A, args.c, args.b_vec.numpy().astype(np.float64)
colptr = A.ccol_indices().numpy()
rows = A.row_indices().numpy()
a = A.values().numpy().astype(np.float64)
c = C.values().numpy().astype(np.float64)
lens = np.diff(colptr)
row_deg = np.bincount(rows, minlength=D)
Q = np.array([0.0, 0.01, 0.05, 0.1, 0.25, 0.5, 0.75, 0.9, 0.95, 0.99, 1.0])
np.savez(
    os.path.join(HERE, "g5_stats_1m.npz"),
    params=np.array([S, D, SP, SEED]),
    nnz=np.int64(a.size),
    col_len_hist=np.bincount(lens, minlength=64)[:64],
    col_len_mean=lens.mean(), col_len_max=lens.max(), empty_cols=np.int64((lens == 0).sum()),
    quantiles=Q,
    row_deg_q=np.quantile(row_deg, Q), row_deg_mean=row_deg.mean(),
    a_q=np.quantile(a, Q), a_mean=a.mean(),
    c_q=np.quantile(c, Q), c_mean=c.mean(), c_at_cap=np.float64((c == -0.5).mean()),
    b_q=np.quantile(b, Q), b_mean=b.mean(), b_sum=b.sum(),
    ratio_q=np.quantile(a / -c, Q),  # s_j of the edge's destination
)
print("nnz", a.size, "mean len", lens.mean(), "max", lens.max(), "empty", (lens == 0).sum(), "b mean", b.mean(), "c at cap", (c == -0.5).mean())
