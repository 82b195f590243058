"""developer tool: oracle/torch_path.py wall time against the torch thread count on this host (synthetic columns of ~10 non-zeros)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle.torch_path import ReferencePathObjective
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, 10_000
rng = np.random.default_rng(0)
lens = rng.poisson(10, n)
colptr = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=colptr[1:])
E = int(colptr[-1])
rows = rng.integers(0, m, E)
c = -np.minimum(rng.lognormal(-4, 0.75, E), 0.5).astype(np.float32); a = (-c * rng.lognormal(0, 1, E)).astype(np.float32)
for th in [int(t) for t in (sys.argv[2:] or ["8", "32", "128"])]:
    torch.set_num_threads(th)
    f = ReferencePathObjective(m, n, colptr, rows, a, c, [("box", {"lower": 0.0, "upper": 1.0}, np.arange(n // 2)), ("simplex", {"z": 1.0}, np.arange(n // 2, n))], 1e-3)
    lam = torch.zeros(m)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); f.calculate(lam); ts.append(time.perf_counter() - t)
    print(f"threads {th}: {n} entities {E} nnz: {min(ts[1:])*1e3:.0f} ms/iteration", flush=True)
