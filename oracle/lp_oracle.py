"""oracle/lp_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the reference's generic-LP ("miplib2017") objective, in the reference's working precision:

  lp_calculate()        z, projected x, dual gradient, objective pieces      objectives/miplib.py:60-109
  bounds_from_map()     per-variable bounds of a projection map             objectives/miplib.py:111-121 (+ box.py:7-16, cone.py:6-28)
  convergence_bound()   PDLP stopping quantities                            objectives/miplib.py:123-230

Pinned against tests/golden/g6_*.npz (tests/test_oracle_golden.py); the AGD loop around it is oracle/agd_oracle.maximize.
"""
import numpy as np


def bounds_from_map(n, entries, dtype):
    """(lower, upper) clamp arrays, -inf / +inf where a bound is absent.  ``entries``: iterable of
    (proj_type, params, indices).  A box bound whose key is missing takes BoxProjection's default (lower=0.0, upper=1.0: box.py:12-13 --
    ``{"upper": 1}`` is [0, 1], tests/test_equality_constraints.py:39) unless the entry is spelled ``l`` / ``u`` (the bound reader's names,
    miplib.py:111-121: a missing key is "no bound"); a key present with NaN / None is an absent bound."""
    lo = np.full(n, -np.inf, dtype=dtype)
    hi = np.full(n, np.inf, dtype=dtype)

    def get(p, *names):
        for k in names:
            if k in p and p[k] is not None and p[k] == p[k]:
                return p[k]
        return None

    for kind, params, idx in entries:
        idx = np.asarray(idx, dtype=np.int64)
        l, u = get(params, "lower", "l"), get(params, "upper", "u")
        if kind == "box":
            short = "l" in params or "u" in params
            lo[idx] = l if l is not None else (-np.inf if (short or "lower" in params) else 0.0)
            hi[idx] = u if u is not None else (np.inf if (short or "upper" in params) else 1.0)
        elif kind == "cone":
            if l is not None:
                lo[idx] = l
            if u is not None:
                hi[idx] = u
        else:
            raise ValueError(f"lp_oracle handles point-wise bounds only, not {kind}")
    return lo, hi


def lp_calculate(A, c, b, lo, hi, lam, gamma, dtype, row_norms=None):
    """miplib.py:60-109 with a dense ``A`` (m x n).  Returns (grad, x, dual_obj, reg, primal_obj) in ``dtype``."""
    T = np.dtype(dtype).type
    A = np.asarray(A, dtype=dtype)
    c, b = np.asarray(c, dtype=dtype), np.asarray(b, dtype=dtype)
    lam = np.asarray(lam, dtype=dtype)
    if row_norms is not None:
        lam = ((T(1) / np.asarray(row_norms, dtype=dtype)) * lam).astype(dtype)     # :74-75
    z = (T(-1.0 / gamma) * ((A.T @ lam).astype(dtype) + c)).astype(dtype)            # :77
    x = np.minimum(np.maximum(z, np.asarray(lo, dtype=dtype)), np.asarray(hi, dtype=dtype)).astype(dtype)  # :80-92
    resid = ((A @ x).astype(dtype) - b).astype(dtype)
    grad = ((T(1) / np.asarray(row_norms, dtype=dtype)) * resid).astype(dtype) if row_norms is not None else resid  # :94-97
    nrm = T(np.sqrt(np.dot(x.astype(np.float64), x.astype(np.float64))))
    reg = T(T(gamma / 2.0) * T(nrm * nrm))                                             # :99
    primal = T(np.dot(c, x))
    obj = T(T(primal + reg) + T(np.dot(lam, resid)))                                   # :101
    return grad, x, obj, reg, primal


def convergence_bound(A, c, b, lower, upper, lam, x=None, optimal_primal_obj=None, tol=1e-4, eq_mask=None, row_norms=None, dtype=np.float64):
    """miplib.py:156-230; ``lower`` / ``upper`` carry NaN where a bound is absent (as :111-121).
    Returns (gap_upperbound, gap_lower_bound, primal_feas, dual_feas, converged)."""
    A = np.asarray(A, dtype=dtype)
    c, b = np.asarray(c, dtype=dtype), np.asarray(b, dtype=dtype)
    lam = np.asarray(lam, dtype=dtype)
    lower, upper = np.asarray(lower, dtype=dtype), np.asarray(upper, dtype=dtype)
    if row_norms is not None:
        lam = (1 / np.asarray(row_norms, dtype=dtype)) * lam
    r = c + A.T @ lam
    if x is None:
        x = np.where(r >= 0, lower, upper)
        if np.isnan(x).any():
            raise ValueError("Unbounded x.")
    x = np.asarray(x, dtype=dtype)
    lam_neg, lam_pos = np.minimum(r, 0), np.maximum(r, 0)
    u_ok, l_ok = ~np.isnan(upper), ~np.isnan(lower)
    d = -np.dot(b, lam) + np.dot(lam_neg[u_ok], upper[u_ok]) + np.dot(lam_pos[l_ok], lower[l_ok])
    p = np.dot(c, x)
    gap_ub = abs(p - d) / (1.0 + abs(p) + abs(d))
    gap_lb = abs(p - optimal_primal_obj) / (1.0 + abs(p) + abs(optimal_primal_obj)) if optimal_primal_obj is not None else float("nan")
    resid = A @ x - b
    viol = np.where(eq_mask, np.abs(resid), np.maximum(resid, 0)) if eq_mask is not None else np.maximum(resid, 0)
    primal_feas = np.linalg.norm(viol) / (1.0 + np.linalg.norm(b))
    xd = -r
    xd = np.where(l_ok & ~u_ok, np.maximum(xd, 0), xd)
    xd = np.where(~l_ok & u_ok, np.minimum(-r, 0), xd)
    xd = np.where(~l_ok & ~u_ok, 0.0, xd)
    dual_feas = np.linalg.norm(r + xd) / (1.0 + np.linalg.norm(c))
    conv = bool(gap_ub <= tol and primal_feas <= tol and dual_feas <= tol)
    return gap_ub, gap_lb, primal_feas, dual_feas, conv
