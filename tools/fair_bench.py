"""Developer tool: cost of the fairness pair at benchmark scale (python tools/fair_bench.py [entities] [proj]):
plain matching objective vs the kernel form (f streamed by the fused kernel) vs the folded form (per-iteration cost rewrite)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from benchmark.synthetic import CHUNK_COLS, generate_matching_problem
from dualip_amd.objectives.matching import MatchingInputArgs, MatchingSolverDualObjectiveFunction
from dualip_amd.objectives.matching_fairness import MatchingFairnessDualObjectiveFunction
from dualip_amd.optimizers.agd import AcceleratedGradientDescent

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
proj = sys.argv[2] if len(sys.argv) > 2 else "simplex"
dev = torch.device("cuda:0")
ranges, pm = bench.shard_plan(proj, n, 1, 0, CHUNK_COLS)
inp = generate_matching_problem(n, 10_000, 1e-3, seed=42, device=dev, dtype=torch.float32, col_ranges=ranges)["input_args"]
inp.projection_map = pm
m = inp.b_vec.numel()
nnz = inp.A.values().numel()


def solve(f, rows, iters=60):
    s = AcceleratedGradientDescent(max_iter=iters, gamma=1e-3, initial_step_size=1e-3, max_step_size=1e-1, iteration_callback=False)
    run = s.start_device_run(f, torch.zeros(rows, dtype=torch.float32, device=dev))
    run.advance(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.advance(iters - 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (iters - 10)
    run.close()
    return dt


t_plain = solve(MatchingSolverDualObjectiveFunction(inp, 1e-3), m)
b2 = torch.cat([inp.b_vec, torch.tensor([1e-4, 1e-4], device=dev)])
args = MatchingInputArgs(A=inp.A, c=inp.c, projection_map=pm, b_vec=b2)
out = {"plain": t_plain}
for native in (True, False):
    f = MatchingFairnessDualObjectiveFunction(args, 1e-3, group_ratio=0.5, native=native)
    out["kernel form" if native else "folded form"] = solve(f, m + 2)
    del f
    torch.cuda.empty_cache()
for k, v in out.items():
    bpn = {"plain": 12, "kernel form": 16, "folded form": 36}[k]
    print(f"{k:12s} {v * 1e3:8.4f} ms/iteration   {nnz * bpn / v / 1e12:5.2f} TB/s at {bpn} B per non-zero   ({v / t_plain:4.2f}x plain)")
