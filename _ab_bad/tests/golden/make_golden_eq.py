#!/usr/bin/env python3
"""Golden vectors of ``simplex_eq`` inside the reference's matching objective (runs ONLY in the build container).

The reference projects every column inside a zero-padded block (one per nnz-bucket with ``batching=True``, one per
entry with ``batching=False``), which changes the result of ``simplex_eq`` whenever a clamped column sums to less than
z.  This script runs the reference's MatchingSolverDualObjectiveFunction on the problem stored in
g1_syn2000.npz for both batching modes and stores the outputs: ``ge_simplex_eq.npz`` (data only).
Re-run with:  python tests/golden/make_golden_eq.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
stub = tempfile.mkdtemp(prefix="mlflow_stub_")
os.makedirs(os.path.join(stub, "mlflow"), exist_ok=True)
open(os.path.join(stub, "mlflow", "__init__.py"), "w").close()
sys.path[:0] = [stub, os.path.join(REF, "src")]

import torch  # noqa: E402
from dualip.objectives.matching import MatchingInputArgs, MatchingSolverDualObjectiveFunction  # noqa: E402
from dualip.projections.base import create_projection_map  # noqa: E402

torch.set_num_threads(4)
z = np.load(os.path.join(HERE, "g1_syn2000.npz"))
m, n = int(z["m"]), int(z["n"])
out = {}
for dn, dt in (("f32", torch.float32), ("f64", torch.float64)):
    colptr, rowidx = torch.from_numpy(z["colptr"]), torch.from_numpy(z["rowidx"])
    A = torch.sparse_csc_tensor(colptr, rowidx, torch.from_numpy(z["a"].copy()).to(dt), size=(m, n))
    C = torch.sparse_csc_tensor(colptr, rowidx, torch.from_numpy(z["c"].copy()).to(dt), size=(m, n))
    b = torch.from_numpy(z["b"].copy()).to(dt)
    for zz in (1.0, 40.0):  # z = 40: most clamped columns sum to less than z at gamma = 0.1 -> the padding matters
        for batching in (True, False):
            pm = create_projection_map("simplex_eq", {"z": zz}, n)
            f = MatchingSolverDualObjectiveFunction(MatchingInputArgs(A=A, c=C, projection_map=pm, b_vec=b, equality_mask=None), gamma=0.1, batching=batching)
            for ln in ("zero", "small"):
                lam = torch.from_numpy(z[f"lam_{ln}"].copy()).to(dt)
                r = f.calculate(lam, gamma=0.1, save_primal=True)
                key = f"{zz}|{int(batching)}|{ln}|{dn}"
                out[f"{key}|grad"] = r.dual_gradient.numpy().copy()
                out[f"{key}|x"] = r.primal_var.numpy().copy()
                out[f"{key}|scal"] = np.array([float(r.dual_objective), float(r.reg_penalty)])
np.savez_compressed(os.path.join(HERE, "ge_simplex_eq.npz"), **out)
x_b1, x_b0 = out["40.0|1|zero|f64|x"], out["40.0|0|zero|f64|x"]
print("keys", len(out), "batching changes x by", np.abs(x_b1 - x_b0).max(), "bytes", os.path.getsize(os.path.join(HERE, "ge_simplex_eq.npz")))
