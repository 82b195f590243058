"""developer tool (build container only -- needs /root/reference): wall time of the reference's own CPU calculate() against
oracle/torch_path.py (the op-sequence restatement bench.py times as CPU baseline) on the same 1M-entity problem."""
import os, sys, tempfile, time
import numpy as np
REF = "/root/reference"
stub = tempfile.mkdtemp(prefix="mlflow_stub_")
os.makedirs(os.path.join(stub, "mlflow"), exist_ok=True)
open(os.path.join(stub, "mlflow", "__init__.py"), "w").close()
sys.path[:0] = [stub, os.path.join(REF, "src"), os.path.join(REF, "benchmark"), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
from generate_synthetic_data import generate_synthetic_matching_input_args
from dualip.objectives.matching import MatchingSolverDualObjectiveFunction
from dualip.projections.base import create_projection_map
from oracle.torch_path import ReferencePathObjective

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "simplex"
args = generate_synthetic_matching_input_args(S, 10_000, 1e-3, device="cpu", dtype=torch.float32, seed=42)
ptype, params = ("simplex", {"z": 1.0}) if kind == "simplex" else ("box", {"lower": 0.0, "upper": 1.0})
args.projection_map = create_projection_map(ptype, params, S)
torch.manual_seed(0)
lam = torch.rand(10_000) * 0.01
f = MatchingSolverDualObjectiveFunction(args, gamma=1e-3)
g = ReferencePathObjective(10_000, S, args.A.ccol_indices().numpy(), args.A.row_indices().numpy(), args.A.values().numpy(), args.c.values().numpy(),
                           [(ptype, params, np.arange(S))], 1e-3)
def best(fn, reps=4):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts[1:]), ts
tr, _ = best(lambda: f.calculate(lam))
to, _ = best(lambda: g.calculate(lam))
r = f.calculate(lam)
ax, o0, sq, x = g.calculate(lam)
print(f"{kind} {S}: reference {tr*1e3:.0f} ms, torch_path {to*1e3:.0f} ms, ratio {to/tr:.2f}; grad max diff {float((r.dual_gradient - (ax - args.b_vec)).abs().max()):.3g}; threads {torch.get_num_threads()}")
