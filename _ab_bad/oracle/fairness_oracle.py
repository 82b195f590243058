"""oracle/fairness_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the matching objective with two fairness rows, the extension the reference documents in
docs/demo/matching_complex.rst (coefficients :45-63, objective :85-166), in the reference's working precision and
operation order.  Small problems only (one projection call per column).

Pinned against tests/golden/gf_fairness.npz (tests/test_oracle_golden.py), which tests/golden/make_golden_fair.py produced
by evaluating the same formulas with the reference's own sparse operators.
"""
import numpy as np

import oracle
from oracle import agd_oracle


def fairness_coefficients(colptr, a, group_ratio, dtype):
    """+a/|T1| on the first int(n * ratio) columns, -a/|T2| on the others (matching_complex.rst:47-61)."""
    T = np.dtype(dtype).type
    n = len(colptr) - 1
    n1 = max(0, min(int(n * group_ratio), n))
    split = int(colptr[n1])
    a = np.asarray(a, dtype=dtype)
    f = np.empty_like(a)
    f[:split] = (T(1 / n1) * a[:split]) if n1 else 0
    f[split:] = (T(-1 / (n - n1)) * a[split:]) if n - n1 else 0
    return f


def fairness_calculate(p, f, lam, gamma, proj, b_full, dtype):
    """Returns (grad - b, dual_obj, reg, primal_obj, x).  ``p``: problem dict (m, n, colptr, rowidx, a, c); ``lam`` and
    ``b_full`` have m + 2 entries; ``proj`` = (proj_type, params) applied to every column."""
    T = np.dtype(dtype).type
    m = p["m"]
    a, c, f = (np.asarray(v, dtype=dtype) for v in (p["a"], p["c"], f))
    rows = np.asarray(p["rowidx"], dtype=np.int64)
    lam = np.asarray(lam, dtype=dtype)
    s = (T(-1.0 / gamma) * lam).astype(dtype)                       # :103
    v = (a * s[:m][rows]).astype(dtype)                             # :106
    v = (v + (s[m] * f).astype(dtype)).astype(dtype)                # :109
    v = (v + (T(T(-1) * s[m + 1]) * f).astype(dtype)).astype(dtype)  # :112
    v = (v + (T(-1.0 / gamma) * c).astype(dtype)).astype(dtype)     # :115
    x = np.zeros_like(v)
    colptr = p["colptr"]
    for j in range(p["n"]):
        k0, k1 = int(colptr[j]), int(colptr[j + 1])
        if k1 > k0:
            x[k0:k1] = oracle.project_dense(v[k0:k1].reshape(-1, 1), proj[0], proj[1]).reshape(-1)  # :118-123
    ax = np.zeros(m + 2, dtype=np.float64)
    np.add.at(ax, rows, (a * x).astype(dtype).astype(np.float64))   # :126
    fx = float(np.sum((f * x).astype(dtype), dtype=np.float64))
    ax[m], ax[m + 1] = fx, -fx                                      # :127-128
    primal = float(np.dot(c.astype(np.float64), x.astype(np.float64)))
    ssq = float(np.dot(x.astype(np.float64), x.astype(np.float64)))
    grad, obj, reg, _, _, _ = agd_oracle.epilogue(ax.astype(dtype), primal, ssq, lam, b_full, gamma, dtype)
    return grad, obj, reg, T(primal), x
