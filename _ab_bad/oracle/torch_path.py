"""oracle/torch_path.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A torch-on-CPU restatement of the reference's OP SEQUENCE for ``calculate`` -- the thing `device="cpu"` runs in the reference:
one gather-scale pass, one add pass, a zero-padded dense [L x K] block per (projection key, nnz bucket) that is filled,
projected with whole-block tensor ops (clamp / sort + cumsum + threshold) and scattered back, one multiply pass, one
scatter-add, a norm and a dot (src/dualip/objectives/matching.py:60-161; utils/sparse_utils.py:26-85, 133-243;
projections/simplex.py:126-236, box.py:15-16, cone.py:21-28).  It is what bench.py times as the CPU baseline beside the C port
(oracle/matching_oracle.c): the C port is a per-column loop and far faster than what the reference executes, this module costs
what the reference's own path costs (SURVEY.md 8d asks for it; checked against the reference's goldens in
tests/test_oracle_golden.py; wall time against the reference itself in the build container: tools/compare_cpu_path.py).

One deliberate difference: every projection key projects its OWN columns -- the reference's multi-key loop
overwrites the other keys' columns with uninitialised memory (sparse_utils.py:177,220).
"""
import numpy as np
import torch


def _thresholds(m):
    th, i = [0], 1
    while 2**i <= m:  # matching.py:93-99
        th.append(2**i)
        i += 1
    th.append(m + 1)
    return th


def _project_block(block, ptype, params):
    """Whole-block projection, one vector per column, zero padding included (as the reference's operators see it)."""
    if ptype == "box":
        return block.clamp(params.get("lower", 0.0), params.get("upper", 1.0))  # box.py:15-16
    if ptype == "cone":
        if params.get("lower") is not None:
            return block.clamp(min=params["lower"])
        if params.get("upper") is not None:
            return block.clamp(max=params["upper"])
        return block
    z = float(params.get("z", 1.0))
    ineq = ptype == "simplex"
    x = block.clamp(min=0.0)  # simplex.py:147
    L, B = x.shape
    out = torch.empty_like(x)
    todo = torch.ones(B, dtype=torch.bool)
    if ineq:  # :153-158
        ok = x.sum(0) <= z + 1e-6
        out[:, ok] = x[:, ok]
        todo = ~ok
    if L > 1 and todo.any():  # top-2 vertex shortcut, :166-193
        idx = todo.nonzero(as_tuple=True)[0]
        top, arg = torch.topk(x[:, idx] / z, 2, dim=0)
        hit = (top[0] - top[1]) > 1.0
        if hit.any():
            cols = idx[hit]
            sol = torch.zeros(L, cols.numel(), dtype=x.dtype)
            sol[arg[0, hit], torch.arange(cols.numel())] = z
            out[:, cols] = sol
            todo[cols] = False
    idx = todo.nonzero(as_tuple=True)[0]
    for c0 in range(0, idx.numel(), 10000):  # :202 chunks of 10 000 columns
        cols = idx[c0 : c0 + 10000]
        sub = x[:, cols]
        u, _ = sub.sort(dim=0, descending=True)
        cs = u.cumsum(0)
        k = torch.arange(1, L + 1, dtype=x.dtype).view(L, 1)
        cond = u - (cs - z) / k > 0
        rho = (cond.to(torch.long) * torch.arange(L).view(L, 1)).max(0).values
        theta = (cs[rho, torch.arange(cols.numel())] - z) / (rho.to(x.dtype) + 1)
        out[:, cols] = (sub - theta.unsqueeze(0)).clamp(min=0)
    return out


class ReferencePathObjective:
    """``calculate`` of the matching objective as the reference's CPU path executes it.  entries: [(ptype, params, column index
    array)] -- every column in at most one entry."""

    def __init__(self, m, n, colptr, rowidx, a, c, entries, gamma, batching=True, dtype=torch.float32):
        self.m, self.n, self.gamma = int(m), int(n), float(gamma)
        self.colptr = torch.as_tensor(np.ascontiguousarray(colptr), dtype=torch.int64)
        self.rows = torch.as_tensor(np.ascontiguousarray(rowidx), dtype=torch.int64)
        self.a = torch.as_tensor(np.ascontiguousarray(a)).to(dtype)
        self.c = torch.as_tensor(np.ascontiguousarray(c)).to(dtype)
        lens = self.colptr[1:] - self.colptr[:-1]
        th = torch.tensor(_thresholds(self.m), dtype=torch.int64)
        bucket_id = torch.bucketize(lens, th)  # matching.py:104
        self.buckets = []  # (ptype, params, [column-index tensors]) -- what the reference keeps (matching.py:70-76)
        for ptype, params, cols in entries:
            cols = torch.as_tensor(np.ascontiguousarray(cols), dtype=torch.int64)
            groups = [cols[bucket_id[cols] == j] for j in range(1, len(th))] if batching else [cols]
            self.buckets.append((ptype, dict(params or {}), [g for g in groups if g.numel() > 0]))

    def calculate(self, lam, gamma=None):
        """Returns (A x [m], c.x, sum x^2, x) like oracle.matching_calculate."""
        if gamma is not None:
            self.gamma = float(gamma)
        lam = torch.as_tensor(lam).to(self.a.dtype)
        scaled = -1.0 / self.gamma * lam                      # matching.py:136
        inter = self.a * scaled[self.rows]                    # left_multiply_sparse
        inter = inter + (-1.0 / self.gamma * self.c)          # elementwise_csc(add); c_rescaled
        x = inter.clone()
        for ptype, params, groups in self.buckets:             # apply_F_to_columns, once per projection key
            for g in groups:                                   # sparse_utils.py:180-214: the block's index vectors are rebuilt per call
                starts = self.colptr[g]
                ln = self.colptr[g + 1] - starts
                total = int(ln.sum().item())
                if total == 0:
                    continue
                L, K = int(ln.max().item()), int(g.numel())
                colpos = torch.arange(K).repeat_interleave(ln)
                offs = torch.arange(total) - (ln.cumsum(0) - ln)[colpos]
                flat = starts[colpos] + offs
                block = torch.zeros((L, K), dtype=inter.dtype)
                block[offs, colpos] = inter[flat]
                x[flat] = _project_block(block, ptype, params)[offs, colpos]
        ax = torch.zeros(self.m, dtype=x.dtype).scatter_add_(0, self.rows, self.a * x)  # row_sums_csc(elementwise_csc(mul))
        return ax, float(torch.dot(self.c, x)), float(torch.norm(x) ** 2), x
