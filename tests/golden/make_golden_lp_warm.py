#!/usr/bin/env python3
"""Golden vectors of BASELINE config 5 as worded -- the generic-LP objective WITH a warm start -- and of the reference's
equality-constraint known answer (fixture G6w; runs ONLY in the build container).

Imports the reference (read-only /root/reference, empty ``mlflow`` stub as make_golden.py) and runs ITS ``run_solver``
(run_solver.py:74-146) with ``objective_type="miplib2017"``:

  g6_lp_warm.npz
    small|{dn}|cold_*    the seeded 40 x 60 LP of g6_lp_small.npz (same arrays, equality rows 3/17/29): 120 iterations from zero duals
    small|{dn}|warm_*    ... then ``SolverArgs(initial_dual_path=<the cold run's dual_val saved with torch.save>)``: 80 more iterations
                         (run_solver.py:127-132: the optimiser restarts -- step-size history, momentum index -- from the loaded duals)
    v150|{dn}|cold_* / warm_*   the shipped MIPLIB instance: 300 iterations, then 200 warm-started ones
    eq2|*                the 2-variable LP of tests/test_equality_constraints.py:18-61 (x1 + x2 = 4, 0 <= x1 <= 1, 0 <= x2;
                         optimum 7.0): the reference's dual objective log over its 1000 iterations and the final dual

Data only; nothing of the reference's source is stored.  Re-run with:  python tests/golden/make_golden_lp_warm.py
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_lp as G  # noqa: E402  (sets up the reference import path and the mlflow stub)

import torch  # noqa: E402
from dualip.objectives.miplib import MIPLIBInputArgs  # noqa: E402
from dualip.projections.base import create_projection_map  # noqa: E402
from dualip.run_solver import run_solver  # noqa: E402
from dualip.types import ComputeArgs, ObjectiveArgs, SolverArgs  # noqa: E402

DT = G.DT


def solve(args, iters, gamma, s0, path=None):
    sa = SolverArgs(max_iter=iters, gamma=gamma, initial_step_size=s0, max_step_size=0.1, save_primal=True, initial_dual_path=path)
    with contextlib.redirect_stdout(io.StringIO()):
        res = run_solver(args, sa, ComputeArgs(host_device="cpu"), ObjectiveArgs(objective_type="miplib2017"))
    return dict(obj_log=np.array(res.dual_objective_log, dtype=np.float64), step_log=np.array(res.step_size_log, dtype=np.float64),
                lam=res.dual_val.numpy().copy(), x=res.objective_result.primal_var.numpy().copy()), res


def cold_then_warm(out, tag, make_args, m, n_cold, n_warm, gamma, s0):
    for dn, dt in DT.items():
        with tempfile.TemporaryDirectory() as td:
            cold, res = solve(make_args(dt), n_cold, gamma, s0)
            path = os.path.join(td, "dual.pt")
            torch.save(res.dual_val, path)
            warm, _ = solve(make_args(dt), n_warm, gamma, s0, path)
        for k, v in cold.items():
            out[f"{tag}|{dn}|cold_{k}"] = v
        for k, v in warm.items():
            out[f"{tag}|{dn}|warm_{k}"] = v
        print(tag, dn, "cold last", cold["obj_log"][-1], "warm first/last", warm["obj_log"][0], warm["obj_log"][-1])
    out[f"{tag}|params"] = np.array([n_cold, n_warm, gamma, s0])


def main():
    out = {}
    p = G.small_problem()

    def small_args(dt):
        return MIPLIBInputArgs(A=torch.from_numpy(p["A"]).to(dt).to_sparse_coo(), c=torch.from_numpy(p["c"]).to(dt), b_vec=torch.from_numpy(p["b"]).to(dt),
                               projection_map=G.small_map(p), equality_mask=torch.from_numpy(p["eq"]))

    cold_then_warm(out, "small", small_args, p["m"], 120, 80, 1e-2, 1e-3)

    data = G.read_mps_file(os.path.join(G.REF, "examples", "miplib_2017", "v150d30-2hopcds.mps.gz"))

    def v150_args(dt):
        d = data.to_dualip_format(dtype=dt)
        return MIPLIBInputArgs(A=d.A, c=d.C, b_vec=d.b_vec, projection_map=d.projection_map, equality_mask=d.equality_mask)

    cold_then_warm(out, "v150", v150_args, len(data.b_vec), 300, 200, 1e-3, 1e-5)

    # the reference's own known answer (tests/test_equality_constraints.py:18-61), through its classes
    from dualip.objectives.miplib import MIPLIB2017ObjectiveFunction
    from dualip.optimizers.agd import AcceleratedGradientDescent

    A = torch.tensor([[1.0, 1.0]])
    args = MIPLIBInputArgs(A=A, c=torch.tensor([1.0, 2.0]), projection_map=create_projection_map("box", {"upper": 1}, num_indices=2, indices=[0]),
                           b_vec=torch.tensor([4.0]), equality_mask=torch.tensor([True]))
    with contextlib.redirect_stdout(io.StringIO()):
        res = AcceleratedGradientDescent(max_iter=1000, gamma=1e-5).maximize(MIPLIB2017ObjectiveFunction(miplib_input_args=args), torch.tensor([0.0]))
    out["eq2|obj_log"] = np.array(res.dual_objective_log, dtype=np.float64)
    out["eq2|step_log"] = np.array(res.step_size_log, dtype=np.float64)
    out["eq2|lam"] = res.dual_val.numpy().copy()
    print("eq2 final dual objective", res.dual_objective, "dual", res.dual_val)
    np.savez_compressed(os.path.join(HERE, "g6_lp_warm.npz"), **out)
    print("g6_lp_warm.npz", os.path.getsize(os.path.join(HERE, "g6_lp_warm.npz")), "bytes")


if __name__ == "__main__":
    main()
