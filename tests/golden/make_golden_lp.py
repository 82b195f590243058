#!/usr/bin/env python3
"""Golden vectors of the generic-LP ("miplib2017") objective (fixture G6; runs ONLY in the build container).

Imports the reference (read-only /root/reference, with the same empty ``mlflow`` stub as make_golden.py), runs ITS
MIPLIB2017ObjectiveFunction / AcceleratedGradientDescent / MPS reader and stores inputs + outputs as ``.npz``:

  g6_miplib_v150.npz   the shipped instance examples/miplib_2017/v150d30-2hopcds.mps.gz as COO arrays + bounds, single
                       calculate() results at two duals and a 2000-iteration AGD trace (fp32 and fp64)
  g6_lp_small.npz      a seeded 40 x 60 LP with every bound shape (two-sided, lower only, upper only, free, unit box
                       defaults), equality rows, dense-A Jacobi preconditioning (a non-point-wise operator on an index set is not
                       captured: the reference fails on its [k, 1] result, miplib.py:90);
                       calculate() results, AGD traces and the PDLP convergence-bound quantities

Data only; nothing of the reference's source is stored.  Re-run with:  python tests/golden/make_golden_lp.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

_stub = tempfile.mkdtemp(prefix="mlflow_stub_")
os.makedirs(os.path.join(_stub, "mlflow"), exist_ok=True)
open(os.path.join(_stub, "mlflow", "__init__.py"), "w").close()
sys.path.insert(0, _stub)
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(REF, "examples", "miplib_2017"))

import torch  # noqa: E402
from dualip.objectives.miplib import MIPLIB2017ObjectiveFunction, MIPLIBInputArgs  # noqa: E402
from dualip.optimizers.agd import AcceleratedGradientDescent  # noqa: E402
from dualip.projections.base import ProjectionEntry  # noqa: E402
from read_mps_data import read_mps_file  # noqa: E402

torch.set_num_threads(4)
DT = {"f32": torch.float32, "f64": torch.float64}


def f(v):
    return float(v.item()) if hasattr(v, "item") else float(v)


def run_calc(obj, lam, gamma):
    r = obj.calculate(lam, gamma, save_primal=True)
    return dict(grad=r.dual_gradient.numpy().copy(), x=r.primal_var.numpy().copy(), scal=np.array([f(r.dual_objective), f(r.reg_penalty), f(r.primal_objective)]))


def run_trace(obj, m, dt, iters, gamma, s0, s1=0.1):
    import contextlib
    import io

    solver = AcceleratedGradientDescent(max_iter=iters, gamma=gamma, initial_step_size=s0, max_step_size=s1, save_primal=True)
    with contextlib.redirect_stdout(io.StringIO()):
        res = solver.maximize(obj, torch.zeros(m, dtype=dt))
    return dict(
        obj_log=np.array(res.dual_objective_log, dtype=np.float64),
        step_log=np.array(res.step_size_log, dtype=np.float64),
        lam=res.dual_val.numpy().copy(),
        x=res.objective_result.primal_var.numpy().copy(),
    )


def put(out, prefix, d):
    for k, v in d.items():
        out[f"{prefix}|{k}"] = v


# ------------------------------------------------------------------------------------------------------------
def make_miplib():
    data = read_mps_file(os.path.join(REF, "examples", "miplib_2017", "v150d30-2hopcds.mps.gz"))
    rows, cols = zip(*data.A_indices)
    out = dict(
        m=np.int64(len(data.b_vec)), n=np.int64(len(data.C_vec)),
        coo_row=np.array(rows, dtype=np.int32), coo_col=np.array(cols, dtype=np.int32), coo_val=np.array(data.A_data, dtype=np.float64),
        c=np.array(data.C_vec, dtype=np.float64), b=np.array(data.b_vec, dtype=np.float64),
        lower=np.array([bd[0] for bd in data.var_bounds], dtype=np.float64), upper=np.array([bd[1] for bd in data.var_bounds], dtype=np.float64),
        equality_mask=np.array(data.equality_mask, dtype=bool),
    )
    rng = np.random.default_rng(7)
    lam_r = rng.uniform(0.0, 0.02, size=int(out["m"]))
    out["lam_rand"] = lam_r
    for dn, dt in DT.items():
        d = data.to_dualip_format(dtype=dt)
        args = MIPLIBInputArgs(A=d.A, c=d.C, b_vec=d.b_vec, projection_map=d.projection_map, equality_mask=d.equality_mask)
        obj = MIPLIB2017ObjectiveFunction(miplib_input_args=args)
        put(out, f"calc|zero|{dn}", run_calc(obj, torch.zeros(len(data.b_vec), dtype=dt), 1e-3))
        put(out, f"calc|rand|{dn}", run_calc(obj, torch.from_numpy(lam_r).to(dt), 1e-3))
        put(out, f"trace|{dn}", run_trace(obj, len(data.b_vec), dt, 2000, 1e-3, 1e-5))
        print("miplib", dn, "obj at 1/100/1000/2000:", [out[f"trace|{dn}|obj_log"][i - 1] for i in (1, 100, 1000, 2000)])
    np.savez_compressed(os.path.join(HERE, "g6_miplib_v150.npz"), **out)


def small_problem(seed=11, m=40, n=60, density=0.2):
    rng = np.random.default_rng(seed)
    mask = rng.random((m, n)) < density
    A = np.where(mask, np.round(rng.normal(0, 1, (m, n)), 3), 0.0)
    A[5, :] = 0.0  # an all-zero row (Jacobi leaves it alone)
    c = np.round(rng.normal(0, 1, n), 3)
    b = np.round(rng.uniform(0.2, 2.0, m), 3)
    eq = np.zeros(m, dtype=bool)
    eq[[3, 17, 29]] = True
    # bound shapes by variable block
    kinds = {}
    kinds["two"] = list(range(0, 15))        # box lower=-0.5 upper=1.5
    kinds["unit"] = list(range(15, 25))      # box, default parameters (0, 1)
    kinds["lo"] = list(range(25, 35))        # cone lower=0
    kinds["up"] = list(range(35, 45))        # cone upper=0.75
    kinds["lu_names"] = list(range(45, 52))  # box given as l / u
    # 52..59: in no entry (free)
    return dict(A=A, c=c, b=b, eq=eq, kinds=kinds, m=m, n=n)


def small_map(p, names=("lower", "upper")):
    k = p["kinds"]
    lo, up = names
    return {
        "two": ProjectionEntry("box", {lo: -0.5, up: 1.5}, indices=k["two"]),
        "unit": ProjectionEntry("box", {}, indices=k["unit"]),
        "lo": ProjectionEntry("cone", {"lower": 0.0}, indices=k["lo"]),
        "up": ProjectionEntry("cone", {"upper": 0.75}, indices=k["up"]),
        "lu": ProjectionEntry("box", {lo: 0.25, up: 2.0}, indices=k["lu_names"]),
    }


def make_small():
    p = small_problem()
    out = dict(A=p["A"], c=p["c"], b=p["b"], eq=p["eq"], m=np.int64(p["m"]), n=np.int64(p["n"]))
    for name, idx in p["kinds"].items():
        out[f"idx_{name}"] = np.array(idx, dtype=np.int64)
    rng = np.random.default_rng(3)
    lam_r = rng.uniform(0, 0.5, p["m"])
    lam_s = rng.normal(0, 0.3, p["m"])  # signed (equality rows)
    out["lam_rand"], out["lam_signed"] = lam_r, lam_s
    for dn, dt in DT.items():
        A_dense = torch.from_numpy(p["A"]).to(dt)
        c, b = torch.from_numpy(p["c"]).to(dt), torch.from_numpy(p["b"]).to(dt)
        eq = torch.from_numpy(p["eq"])
        for form in ("dense", "coo"):
            A = A_dense if form == "dense" else A_dense.to_sparse_coo()
            args = MIPLIBInputArgs(A=A, c=c, b_vec=b, projection_map=small_map(p), equality_mask=eq)
            obj = MIPLIB2017ObjectiveFunction(miplib_input_args=args)
            for ln, lam in (("zero", np.zeros(p["m"])), ("rand", lam_r), ("signed", lam_s)):
                for g in (1e-2, 0.5):
                    put(out, f"calc|{form}|{ln}|{g}|{dn}", run_calc(obj, torch.from_numpy(lam).to(dt), g))
            if form == "coo":
                put(out, f"trace|plain|{dn}", run_trace(obj, p["m"], dt, 300, 1e-2, 1e-3))
        # Jacobi (dense only in the reference)
        args = MIPLIBInputArgs(A=A_dense, c=c, b_vec=b, projection_map=small_map(p), equality_mask=eq)
        obj = MIPLIB2017ObjectiveFunction(miplib_input_args=args, use_jacobi_precondition=True)
        out[f"row_norms|{dn}"] = obj.row_norms.numpy().copy()
        put(out, f"calc|jacobi|rand|0.01|{dn}", run_calc(obj, torch.from_numpy(lam_r).to(dt), 1e-2))
        put(out, f"trace|jacobi|{dn}", run_trace(obj, p["m"], dt, 300, 1e-2, 1e-3))
        # convergence bound: bounds spelled l / u (what the reference's bound reader understands, miplib.py:117-120)
        pm_lu = {k: ProjectionEntry(e.proj_type, ({"l": e.proj_params.get("lower", float("nan")), "u": e.proj_params.get("upper", float("nan"))}
                                                  if e.proj_params else {"l": 0.0, "u": 1.0}), indices=e.indices)
                 for k, e in small_map(p).items()}
        # variables in no entry are unbounded: give x explicitly there
        args = MIPLIBInputArgs(A=A_dense, c=c, b_vec=b, projection_map=pm_lu, equality_mask=eq)
        obj = MIPLIB2017ObjectiveFunction(miplib_input_args=args)
        xs = torch.from_numpy(np.clip(np.random.default_rng(5).normal(0, 1, p["n"]), -0.5, 0.75)).to(dt)
        for ln, lam in (("rand", lam_r), ("signed", lam_s)):
            gu, gl, pf, df, conv = obj.calculate_convergence_bound(torch.from_numpy(lam).to(dt), x=xs, optimal_primal_obj=torch.tensor(-1.25, dtype=dt), tol=1e-2)
            out[f"bound|{ln}|{dn}"] = np.array([f(gu), f(gl), f(pf), f(df), float(bool(conv))])
        out[f"bound_x|{dn}"] = xs.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g6_lp_small.npz"), **out)


if __name__ == "__main__":  # (make_golden_lp_warm.py imports this module for its helpers)
    make_small()
    make_miplib()
    for name in ("g6_lp_small.npz", "g6_miplib_v150.npz"):
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")
