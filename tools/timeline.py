"""Developer tool: per-workgroup timeline of one fused launch (needs DUALIP_HIP_TIMELINE=1).
usage: DUALIP_HIP_TIMELINE=1 python tools/timeline.py [entities] [proj]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DUALIP_HIP_TIMELINE", "1")
import numpy as np
import torch

import bench
from benchmark.synthetic import CHUNK_COLS, generate_matching_problem
from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
proj = sys.argv[2] if len(sys.argv) > 2 else "mixed"
dev = torch.device("cuda:0")
ranges, pm = bench.shard_plan(proj, n, 1, 0, CHUNK_COLS)
prob = generate_matching_problem(n, 10_000, 1e-3, seed=42, device=dev, dtype=torch.float32, col_ranges=ranges)
inp = prob["input_args"]
inp.projection_map = pm
f = MatchingSolverDualObjectiveFunction(inp, 1e-3)
from dualip_amd.optimizers.agd import AcceleratedGradientDescent

iters = int(sys.argv[3]) if len(sys.argv) > 3 else 25
solver = AcceleratedGradientDescent(max_iter=iters, gamma=1e-3, initial_step_size=1e-3, max_step_size=1e-1, iteration_callback=False)
run = solver.start_device_run(f, torch.zeros(10_000, dtype=torch.float32, device=dev), rank=0)
run.advance(iters)
torch.cuda.synchronize()
raw = f.timeline().astype(np.uint64)
raw[:, 0] &= np.uint64(0x0FFFFFFFFFFFFFFF)  # (top four bits: the workgroup's real XCD)
tl = raw.astype(np.int64)
t0 = tl[:, 0].min()
us = (tl - t0) / 100.0  # 100 MHz
print("info", f.info())
print("balance table", [int(f._lib.dl_matching_info(f._handle, 18 + i)) for i in range(16)])
print("kernel span us", us[:, 3].max())
for k, name in enumerate(["start", "prologue_done", "loop_done", "end"]):
    print(f"{name:14s} min {us[:, k].min():8.1f} mean {us[:, k].mean():8.1f} max {us[:, k].max():8.1f}")
for k, name in ((4, "step derived (fused apply)"), (5, "dual rows staged")):
    col = tl[:, k]
    if (col > 0).any():
        v = (col[col > 0] - t0) / 100.0
        print(f"{name:34s} min {v.min():8.1f} mean {v.mean():8.1f} max {v.max():8.1f}   (workgroup 0: {(col[0] - t0) / 100.0 if col[0] > 0 else float('nan'):.1f})")
print("workgroup 0: prologue done %.1f loop done %.1f end %.1f" % (us[0, 1], us[0, 2], us[0, 3]))
d = us[:, 2] - us[:, 1]
print("loop dur  min %.1f mean %.1f max %.1f" % (d.min(), d.mean(), d.max()))
print("prologue dur mean %.1f max %.1f; epilogue dur mean %.1f max %.1f" % ((us[:, 1] - us[:, 0]).mean(), (us[:, 1] - us[:, 0]).max(), (us[:, 3] - us[:, 2]).mean(), (us[:, 3] - us[:, 2]).max()))
order = np.argsort(us[:, 2])
print("earliest-finishing WGs", order[:8], us[order[:8], 2])
print("latest-finishing WGs", order[-8:], us[order[-8:], 2])
q = np.linspace(0, len(d) - 1, 17).astype(int)
print("loop-done by wg (every 16th):", np.round(us[q, 2], 0))
xcd = np.arange(len(us)) % 8
print("loop-done mean by XCD (us):", np.round([us[xcd == k, 2].mean() for k in range(8)], 1), " overall mean %.1f max %.1f" % (us[:, 2].mean(), us[:, 2].max()))
