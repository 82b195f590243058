"""Developer tool: how often is the simplex support of a whole 256-element window unchanged between consecutive iterations?
(decides whether a warm-started fixed-point check can replace the cold projection)   usage: python tools/support_stability.py [entities] [proj]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from benchmark.synthetic import CHUNK_COLS, generate_matching_problem
from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
from dualip_amd.optimizers.agd import AcceleratedGradientDescent

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
proj = sys.argv[2] if len(sys.argv) > 2 else "simplex"
decay = len(sys.argv) > 3 and sys.argv[3] == "decay"
dev = torch.device("cuda:0")
ranges, pm = bench.shard_plan(proj, n, 1, 0, CHUNK_COLS)
inp = generate_matching_problem(n, 10_000, 1e-3, seed=42, device=dev, dtype=torch.float32, col_ranges=ranges)["input_args"]
inp.projection_map = pm
g0 = 1e-3 / (0.7 ** (1000 // 35)) if decay else 1e-3
f = MatchingSolverDualObjectiveFunction(inp, g0)
solver = AcceleratedGradientDescent(max_iter=1000, gamma=g0, initial_step_size=1e-3, max_step_size=1e-1, iteration_callback=False,
                                    gamma_decay_type="step" if decay else None, gamma_decay_params={"decay_steps": 35, "decay_factor": 0.7} if decay else None)
run = solver.start_device_run(f, torch.zeros(10_000, dtype=torch.float32, device=dev))
probe = torch.empty_like(inp.A.values())
nnz = probe.numel()
w = nnz // 256 * 256
for stop in (5, 10, 20, 35, 100, 300, 800):
    run.advance(stop - run.done)
    lam0 = run._fetch(0).clone()
    g = run.gamma.value
    run.advance(1)
    lam1 = run._fetch(0).clone()
    g1 = run.gamma.value
    x0 = f.calculate(lam0, g, save_primal=True).primal_var
    s0 = (x0 > 0)[:w].view(-1, 256).clone()
    x1 = f.calculate(lam1, g1, save_primal=True).primal_var
    s1 = (x1 > 0)[:w].view(-1, 256)
    same = (s0 == s1).all(dim=1).float().mean().item()
    elem = (s0 != s1).float().mean().item()
    print(f"iteration {stop:4d} -> {stop + 1}: windows with an unchanged support {same * 100:5.1f} %   elements that changed {elem * 100:.4f} %   support density {s1.float().mean().item() * 100:.2f} %")
run.close()
