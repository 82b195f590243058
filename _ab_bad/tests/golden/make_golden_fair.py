#!/usr/bin/env python3
"""Golden vectors of the matching LP with a two-group fairness constraint (fixture GF; runs ONLY in the build container).

The reference ships this extension as a worked example in its documentation (docs/demo/matching_complex.rst), not as code.
This script evaluates the example's formulas WITH THE REFERENCE'S OWN operators imported from /root/reference
(left_multiply_sparse, elementwise_csc, apply_F_to_columns, row_sums_csc, split_csc_by_cols, hstack_csc, the matching
objective's buckets, the projection registry and the AGD maximiser) on the G1 problem, and stores inputs + outputs:

  gf_fairness.npz   group_ratio, delta, the fairness coefficients; calculate() at three duals and a 60-iteration AGD trace,
                    for a simplex map and a box map, fp32 and fp64

Data only.  Re-run with:  python tests/golden/make_golden_fair.py
"""
import contextlib
import io
import os
import sys
import tempfile
from operator import add, mul

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
_stub = tempfile.mkdtemp(prefix="mlflow_stub_")
os.makedirs(os.path.join(_stub, "mlflow"), exist_ok=True)
open(os.path.join(_stub, "mlflow", "__init__.py"), "w").close()
sys.path.insert(0, _stub)
sys.path.insert(0, os.path.join(REF, "src"))

import torch  # noqa: E402
from dualip.objectives.matching import MatchingInputArgs, MatchingSolverDualObjectiveFunction  # noqa: E402
from dualip.optimizers.agd import AcceleratedGradientDescent  # noqa: E402
from dualip.projections.base import create_projection_map, project  # noqa: E402
from dualip.types import ObjectiveResult  # noqa: E402
from dualip.utils.sparse_utils import apply_F_to_columns, elementwise_csc, hstack_csc, left_multiply_sparse, row_sums_csc, split_csc_by_cols  # noqa: E402

torch.set_num_threads(4)
DT = {"f32": torch.float32, "f64": torch.float64}
GROUP_RATIO, DELTA = 0.4, 0.0


class WithFairnessRows(MatchingSolverDualObjectiveFunction):
    """The matching objective with the two extra rows of the documentation's example."""

    def __init__(self, args, gamma, group_ratio):
        self.b_full = args.b_vec
        inner = MatchingInputArgs(A=args.A, c=args.c, projection_map=args.projection_map, b_vec=args.b_vec[:-2], equality_mask=None)
        super().__init__(matching_input_args=inner, gamma=gamma)
        n = self.A.size(1)
        n1 = max(0, min(int(n * group_ratio), n))
        g1, g2 = split_csc_by_cols(self.A, [n1, n - n1])
        self.A_fairness = hstack_csc([1 / n1 * g1, -1 / (n - n1) * g2])

    def calculate(self, dual_val, gamma=None, save_primal=False, **kwargs):
        if gamma is not None and gamma != self.gamma:
            self.gamma = gamma
            self.c_rescaled = -1.0 / gamma * self.c
        s = -1.0 / self.gamma * dual_val
        work = self.intermediate
        left_multiply_sparse(s[:-2], self.A, output_tensor=work)
        elementwise_csc(work, s[-2] * self.A_fairness, add, output_tensor=work)
        elementwise_csc(work, -1 * s[-1] * self.A_fairness, add, output_tensor=work)
        elementwise_csc(work, self.c_rescaled, add, output_tensor=work)
        for buckets, kind, params in self.buckets.values():
            apply_F_to_columns(work, project(kind, **params), buckets, output_tensor=work)
        x = work.values()
        grad = torch.zeros_like(dual_val)
        grad[:-2] = row_sums_csc(elementwise_csc(self.A, work, mul))
        grad[-2] = elementwise_csc(self.A_fairness, work, mul).values().sum()
        grad[-1] = elementwise_csc(-1 * self.A_fairness, work, mul).values().sum()
        reg = (self.gamma / 2) * torch.norm(x) ** 2
        primal = torch.dot(self.c.values(), x)
        grad = grad - self.b_full
        obj = primal + reg + torch.dot(dual_val, grad)
        res = ObjectiveResult(dual_gradient=grad, dual_objective=obj, reg_penalty=reg, dual_val_times_grad=torch.dot(dual_val, grad),
                              max_pos_slack=max(torch.max(grad), 0), sum_pos_slack=torch.relu(grad).sum())
        if save_primal:
            res.primal_var, res.primal_objective = x.clone(), primal.clone()
        return res


def main():
    z = np.load(os.path.join(HERE, "g1_syn2000.npz"))
    m, n = int(z["m"]), int(z["n"])
    out = dict(group_ratio=np.float64(GROUP_RATIO), delta=np.float64(DELTA))
    rng = np.random.default_rng(21)
    lams = {"zero": np.zeros(m + 2), "rand": np.concatenate([rng.uniform(0, 0.01, m), [0.03, 0.0]]), "tilt": np.concatenate([rng.uniform(0, 0.01, m), [0.0, 0.08]])}
    for k, v in lams.items():
        out[f"lam_{k}"] = v
    maps = {"simplex1": ("simplex", {"z": 1.0}), "box01": ("box", {"lower": 0.0, "upper": 1.0})}
    for dn, dt in DT.items():
        colptr, rowidx = torch.from_numpy(z["colptr"]), torch.from_numpy(z["rowidx"])
        A = torch.sparse_csc_tensor(colptr, rowidx, torch.from_numpy(z["a"]).to(dt), size=(m, n))
        C = torch.sparse_csc_tensor(colptr, rowidx, torch.from_numpy(z["c"]).to(dt), size=(m, n))
        b = torch.cat([torch.from_numpy(z["b"]).to(dt), torch.tensor([DELTA, DELTA], dtype=dt)])
        for mn, (kind, params) in maps.items():
            args = MatchingInputArgs(A=A, c=C, projection_map=create_projection_map(kind, params, n), b_vec=b)
            obj = WithFairnessRows(args, 0.02, GROUP_RATIO)
            out[f"f|{dn}"] = obj.A_fairness.values().numpy().copy()
            for ln, lam in lams.items():
                r = obj.calculate(torch.from_numpy(lam).to(dt), 0.02, save_primal=True)
                pre = f"calc|{mn}|{ln}|{dn}"
                out[pre + "|grad"] = r.dual_gradient.numpy().copy()
                out[pre + "|x"] = r.primal_var.numpy().copy()
                out[pre + "|scal"] = np.array([float(r.dual_objective), float(r.reg_penalty), float(r.primal_objective)])
            solver = AcceleratedGradientDescent(max_iter=60, gamma=0.02, initial_step_size=1e-3, max_step_size=0.1, save_primal=True)
            with contextlib.redirect_stdout(io.StringIO()):
                res = solver.maximize(WithFairnessRows(args, 0.02, GROUP_RATIO), torch.zeros(m + 2, dtype=dt))
            pre = f"trace|{mn}|{dn}"
            out[pre + "|obj_log"] = np.array(res.dual_objective_log)
            out[pre + "|step_log"] = np.array(res.step_size_log)
            out[pre + "|lam"] = res.dual_val.numpy().copy()
            out[pre + "|x"] = res.objective_result.primal_var.numpy().copy()
            print(mn, dn, "obj at 1/30/60:", [res.dual_objective_log[i - 1] for i in (1, 30, 60)], "fair duals", res.dual_val[-2:].tolist())
    path = os.path.join(HERE, "gf_fairness.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
