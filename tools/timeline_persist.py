"""Developer tool: are the workgroups' finish-time deviations persistent from launch to launch?  (needs DUALIP_HIP_TIMELINE=1)
usage: DUALIP_HIP_TIMELINE=1 python tools/timeline_persist.py [entities] [proj]"""
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
from dualip_amd.optimizers.agd import AcceleratedGradientDescent

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
proj = sys.argv[2] if len(sys.argv) > 2 else "mixed"
dev = torch.device("cuda:0")
ranges, pm = bench.shard_plan(proj, n, 1, 0, CHUNK_COLS)
prob = generate_matching_problem(n, 10_000, 1e-3, seed=42, device=dev, dtype=torch.float32, col_ranges=ranges)
inp = prob["input_args"]
inp.projection_map = pm
f = MatchingSolverDualObjectiveFunction(inp, 1e-3)
solver = AcceleratedGradientDescent(max_iter=200, gamma=1e-3, initial_step_size=1e-3, max_step_size=1e-1, iteration_callback=False)
run = solver.start_device_run(f, torch.zeros(10_000, dtype=torch.float32, device=dev), rank=0)
res = []
for stop in (40, 41, 42, 60, 61):
    run.advance(stop)
    torch.cuda.synchronize()
    raw = f.timeline().astype(np.uint64)
    xcc = (raw[:, 0] >> np.uint64(60)).astype(np.int64)
    raw[:, 0] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    tl = raw.astype(np.int64)
    print('   real XCD of workgroups 0..15:', xcc[:16].tolist(), ' wg%8 matches for', int((xcc == np.arange(len(xcc)) % 8).sum()), 'of', len(xcc))
    us = (tl - tl[:, 0].min()) / 100.0
    d = us[:, 2]
    xcd = np.arange(len(d)) % 8
    r = d - np.array([d[xcd == k].mean() for k in range(8)])[xcd]  # deviation from the XCD's mean
    res.append((stop, d.copy(), r))
    print("launch %d: span %.1f mean %.1f max %.1f  within-XCD residual std %.2f us max %.2f" % (stop, us[:, 3].max(), d.mean(), d.max(), r.std(), r.max()))
for a in range(len(res)):
    for b in range(a + 1, len(res)):
        print("residual correlation launches %d / %d: %.3f   (whole finish times: %.3f)" % (res[a][0], res[b][0], np.corrcoef(res[a][2], res[b][2])[0, 1], np.corrcoef(res[a][1], res[b][1])[0, 1]))
print("info", f.info())
