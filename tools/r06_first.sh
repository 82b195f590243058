#!/bin/bash
# round 6, first GPU call: the suite at HEAD, config 2 diagnostics (32-bit slab overflows, timeline), the 12.5M shard timeline
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 ) > gpurun_out/r06a_suite.txt 2>&1
tail -5 gpurun_out/r06a_suite.txt
python bench.py --entities 1000000 --proj box --steps 10 --warmup 3 --no-cpu-baseline --no-late --no-verify 2>/dev/null | tail -1 > gpurun_out/r06a_cfg2_line.json
python - <<'PY' > gpurun_out/r06a_cfg2_overflows.txt 2>&1
import torch, sys
sys.path.insert(0, ".")
import bench
from benchmark.synthetic import CHUNK_COLS, generate_matching_problem
from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
from dualip_amd.optimizers.agd import AcceleratedGradientDescent
dev = torch.device("cuda:0")
for n, proj in ((1_000_000, "box"), (10_000_000, "mixed")):
    ranges, pm = bench.shard_plan(proj, n, 1, 0, CHUNK_COLS)
    prob = generate_matching_problem(n, 10_000, 1e-3, seed=42, device=dev, dtype=torch.float32, col_ranges=ranges)
    inp = prob["input_args"]; inp.projection_map = pm
    f = MatchingSolverDualObjectiveFunction(inp, 1e-3)
    solver = AcceleratedGradientDescent(max_iter=200, gamma=1e-3, initial_step_size=1e-3, max_step_size=1e-1, iteration_callback=False)
    run = solver.start_device_run(f, torch.zeros(10_000, dtype=torch.float32, device=dev), rank=0)
    for k in (1, 1, 1, 5, 8, 16, 32, 64):
        run.advance(k); torch.cuda.synchronize()
        print(n, proj, "after +", k, "overflowing workgroups", f.info()["slab_overflows"], "slab bytes", f.info()["slab_bytes"])
PY
cat gpurun_out/r06a_cfg2_overflows.txt
DUALIP_HIP_TIMELINE=1 python tools/timeline.py 1000000 box > gpurun_out/r06a_timeline_1m_box.txt 2>&1
DUALIP_HIP_TIMELINE=1 python tools/timeline.py 12500000 mixed > gpurun_out/r06a_timeline_12m5_mixed.txt 2>&1
tail -12 gpurun_out/r06a_timeline_1m_box.txt; tail -12 gpurun_out/r06a_timeline_12m5_mixed.txt
