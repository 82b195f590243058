"""Developer tool: per-workgroup timeline of one fused launch on the MovieLens-shaped problem (benchmark/movielens_like.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DUALIP_HIP_TIMELINE", "1")
import numpy as np
import torch

from benchmark.movielens_like import generate
from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
from dualip_amd.optimizers.agd import AcceleratedGradientDescent

from dualip_amd.objectives.matching import MatchingInputArgs
from dualip_amd.projections import create_projection_map

A, C, counts = generate()
m, n = int(A.shape[0]), int(A.shape[1])
inp = MatchingInputArgs(A=A, c=C, projection_map=create_projection_map("simplex", {"z": 1.0}, n, indices=range(n)), b_vec=torch.full((m,), 30.0, device="cuda:0"))
f = MatchingSolverDualObjectiveFunction(inp, 0.1)
stress = "--stress" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
iters = int(argv[0]) if argv else 50
solver = AcceleratedGradientDescent(max_iter=iters, gamma=0.1, initial_step_size=1e-8, max_step_size=1e-6, iteration_callback=False)
from benchmark.movielens_like import stress_duals

run = solver.start_device_run(f, stress_duals(m, "cuda:0") if stress else torch.zeros(m, dtype=torch.float32, device="cuda:0"), rank=0)
run.advance(iters)
torch.cuda.synchronize()
tl = f.timeline().astype(np.int64) & 0x0FFFFFFFFFFFFFFF  # (the top four bits of stamp 0 carry the XCD id)
us = (tl - tl[:, 0].min()) / 100.0
print("info", f.info())
for k, name in enumerate(["start", "stamp1", "loop_done", "end"]):
    print(f"{name:10s} min {us[:, k].min():8.1f} mean {us[:, k].mean():8.1f} max {us[:, k].max():8.1f}")
print("phase durations (mean / max): 0->1 %.1f / %.1f   1->2 %.1f / %.1f   2->3 %.1f / %.1f" % (
    (us[:, 1] - us[:, 0]).mean(), (us[:, 1] - us[:, 0]).max(), (us[:, 2] - us[:, 1]).mean(), (us[:, 2] - us[:, 1]).max(), (us[:, 3] - us[:, 2]).mean(), (us[:, 3] - us[:, 2]).max()))
end = us[:, 3]
order = np.argsort(-end)
print("percentiles of end: 50 %% %.1f  90 %% %.1f  99 %% %.1f  max %.1f" % tuple(np.percentile(end, [50, 90, 99, 100])))
print("latest workgroups (index: stamp1 / loop_done / end):", "  ".join("%d: %.1f / %.1f / %.1f" % (w, us[w, 1], us[w, 2], us[w, 3]) for w in order[:12]))
