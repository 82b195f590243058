"""Randomised parity sweep (``-m gpu``): many small seeded problems with random shapes, degree laws, TIED values (ratings-like
discrete levels, which stress the Newton passes' termination), random projection maps and duals -- HIP pass vs the CPU oracle.
Tolerance: RTOL of tests/helpers.py."""
import numpy as np
import pytest
import torch

import oracle
from oracle import agd_oracle
from tests.helpers import NP_DT, RTOL, relerr, torch_args

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(seed):
    rng = np.random.default_rng(seed)
    m = int(rng.choice([7, 40, 300, 2500]))
    n = int(rng.choice([1, 5, 60, 900, 4000]))
    law = rng.choice(["poisson", "heavy", "dense"])
    if law == "poisson":
        deg = rng.poisson(rng.choice([2, 9, 30]), n)
    elif law == "heavy":
        deg = np.minimum((rng.pareto(1.2, n) * 4).astype(np.int64), 2000)
    else:
        deg = np.full(n, m)
    deg = np.minimum(deg, m).astype(np.int64)
    if n > 3:
        deg[rng.integers(0, n, size=max(1, n // 20))] = 0
    colptr = np.zeros(n + 1, dtype=np.int64)
    colptr[1:] = np.cumsum(deg)
    rows = np.concatenate([np.sort(rng.choice(m, size=int(d), replace=False)) for d in deg]) if deg.sum() else np.zeros(0, dtype=np.int64)
    nnz = rows.size
    tied = rng.random() < 0.5
    a = np.ones(nnz) if tied else rng.uniform(0.05, 1.0, nnz)
    c = -(rng.integers(1, 11, nnz) * 0.5) if tied else -rng.uniform(0.01, 0.5, nnz)
    p = dict(m=m, n=n, colptr=colptr, rowidx=rows.astype(np.int64), a=a, c=c, b=rng.uniform(0.5, 30.0, m))
    # random map: 1-4 contiguous blocks with random operators, possibly an uncovered gap
    cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=rng.integers(0, 4))]))
    kinds = [("simplex", {"z": float(rng.choice([0.3, 1.0, 2.5]))}), ("box", {"lower": 0.0, "upper": float(rng.choice([0.2, 1.0]))}),
             ("cone", {"lower": 0.0}), ("cone", {"upper": float(rng.choice([0.1, 5.0]))}), None]
    from dualip_amd.projections.base import ProjectionEntry

    pm, entries, col_proj = {}, [], np.full(n, -1, dtype=np.int32)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        k = kinds[int(rng.integers(0, len(kinds)))]
        if k is None or hi == lo:
            continue
        pm[f"e{len(entries)}"] = ProjectionEntry(k[0], dict(k[1]), indices=list(range(lo, hi)))
        col_proj[lo:hi] = len(entries)
        entries.append(k)
    if not entries:
        pm["e0"] = ProjectionEntry("simplex", {"z": 1.0}, indices=list(range(n)))
        entries.append(("simplex", {"z": 1.0}))
        col_proj[:] = 0
    gamma = float(rng.choice([1e-3, 0.05, 0.1, 1.0]))
    lam = rng.uniform(0, rng.choice([0.0, 0.01, 1.0]), m)
    dn = "f32" if rng.random() < 0.5 else "f64"
    return p, pm, entries, col_proj, gamma, lam, dn


@pytest.mark.parametrize("block", range(4))
def test_random_problems_match_the_oracle(block, monkeypatch):
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction

    if block == 3:
        monkeypatch.setenv("DUALIP_HIP_HOT_ROWS", "3")  # hot-rows plan wherever it applies (m > 3, 256-wide layout)
    for seed in range(block * 12, block * 12 + 12):
        p, pm, entries, col_proj, gamma, lam, dn = _case(1000 + seed)
        f = MatchingSolverDualObjectiveFunction(torch_args(p, dn, pm, DEV), gamma=gamma)
        td = torch.float32 if dn == "f32" else torch.float64
        res = f.calculate(torch.from_numpy(lam).to(td).to(DEV), save_primal=True)
        ax, obj0, ssq, x = oracle.matching_calculate(p["m"], p["n"], p["colptr"], p["rowidx"], p["a"], p["c"], lam, gamma, entries, col_proj=col_proj, dtype=NP_DT[dn])
        grad, obj, reg, *_ = agd_oracle.epilogue(ax, obj0, ssq, lam, p["b"], gamma, NP_DT[dn])
        tag = (seed, dn, p["m"], p["n"], int(p["colptr"][-1]), f.info()["layout"], f.info()["hot_rows"])
        assert relerr(res.primal_var.cpu().numpy(), x) < RTOL[dn] * 2, tag
        assert relerr(res.dual_gradient.cpu().numpy(), grad) < RTOL[dn] * 2, tag
        assert relerr([float(res.dual_objective), float(res.reg_penalty)], [obj, reg]) < RTOL[dn] * 20, tag
