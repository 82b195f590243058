#!/usr/bin/env python3
"""BASELINE config 1 at full size, on the GPU: a MovieLens-20M-SHAPED matching problem (the data set itself is not in the
reference tree and there is no network).  Shape and parameters as examples/movielens_matching/movies_lens_matching.py:
138 493 users (columns) x 26 744 movies (rows), ~20 M ratings, a = 1, c = -rating in {0.5, ..., 5}, one simplex z = 1 per
user, capacity b = 30 per movie, gamma = 0.1, step sizes 1e-8 / 1e-6.  Ratings per user are heavy tailed (20 ... ~9 000, mean
~144) and movie popularity is Zipf-like, as in ml-20m -- so most non-zeros sit in columns longer than one 256-element window
(the single-column path) and the dual vector does not fit the LDS (the hot-rows plan).

    python benchmark/movielens_like.py [--max-iter 1000]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


def generate(n_users=138_493, n_movies=26_744, seed=7, device="cuda:0"):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    # ratings per user: log-normal body, clipped to [20, 9254] (ml-20m keeps users with >= 20 ratings)
    deg = torch.exp(torch.randn(n_users, device=device, generator=g) * 1.0 + 4.35).clamp_(20, 9254).to(torch.int64)
    total = int(deg.sum())
    col = torch.repeat_interleave(torch.arange(n_users, device=device), deg, output_size=total)
    # movie popularity ~ rank^-1
    w = 1.0 / torch.arange(1, n_movies + 1, device=device, dtype=torch.float64)
    cdf = torch.cumsum(w / w.sum(), 0)
    mov = torch.searchsorted(cdf, torch.rand(total, device=device, dtype=torch.float64, generator=g)).clamp_(max=n_movies - 1)
    key = torch.unique(col * n_movies + mov)  # sorted by (user, movie), duplicates dropped
    col = torch.div(key, n_movies, rounding_mode="floor")
    mov = key - col * n_movies
    rating = (torch.randint(1, 11, (key.numel(),), device=device, generator=g).float()) * 0.5
    counts = torch.bincount(col, minlength=n_users)
    colptr = torch.zeros(n_users + 1, dtype=torch.int64, device=device)
    colptr[1:] = torch.cumsum(counts, 0)
    A = torch.sparse_csc_tensor(colptr, mov, torch.ones_like(rating), size=(n_movies, n_users), check_invariants=False)
    C = torch.sparse_csc_tensor(colptr, mov, -rating, size=(n_movies, n_users), check_invariants=False)
    return A, C, counts


LENGTH_CLASSES = [(1, 24), (25, 255), (256, 512), (513, 1024), (1025, 1 << 30)]  # one kernel plan each


def stress_duals(m, device, seed=11):
    """A dual vector that makes the projection work: prices of the order of a rating step on the popular movies (row index =
    popularity rank in generate()), so thresholds fall between tied ratings and the Newton passes multiply."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lam = torch.rand(m, device=device, generator=g) * 3.0 / (1.0 + torch.arange(m, device=device) / 2000.0)
    return lam.float()


def verify(f, inp, gamma, lam, device):
    """benchmark/verify.verify_at_size (the checker bench.py runs at the benchmark size) on this problem, at the solve's duals and at
    a stress dual vector: the oracle on slabs of columns -- incl. one slab around a column of every length class -- A x, c.x and
    sum x^2 recomputed in float64 from the primal, the two-handle sharded route against the single objective."""
    from benchmark.verify import verify_at_size

    ok = True
    for tag, lv in (("solve's duals", lam), ("stress duals", stress_duals(lam.numel(), device))):
        out = verify_at_size("f32", gamma, inp, inp.projection_map, f, f, lv.contiguous(), device=device, length_classes=LENGTH_CLASSES)
        worst = max(out["checks"], key=lambda c: c["err"] / max(c["tol"], 1e-300))
        print(f"verify ({tag}): {'ok' if out['ok'] else 'FAILED'}, {len(out['checks'])} checks, worst: {worst['name']} err {worst['err']:.3g} (tol {worst['tol']:g})")
        for c in out["checks"]:
            if not c["ok"]:
                print("   FAILED", c)
        ok = ok and out["ok"]
    if not ok:
        raise SystemExit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-iter", type=int, default=1000)
    ap.add_argument("--capacity", type=float, default=30.0)
    ap.add_argument("--gamma", type=float, default=0.1)
    ap.add_argument("--stress", action="store_true", help="start from stress_duals() instead of zero: the kernel's cost with prices that cut between tied ratings")
    ap.add_argument("--measure-traffic", metavar="OUT.json", default=None, help="also run this script (20 iterations) under rocprofv3 --pmc, one counter per pass -- FETCH_SIZE, "
                    "WRITE_SIZE, SQ_INSTS_VALU, SQ_INSTS_LDS, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAIT_INST_ANY -- and write the fused kernel's per-launch means to OUT.json")
    ap.add_argument("--no-verify", action="store_true", help="skip the checks against the CPU oracle after the solve (they are outside every timed region)")
    args = ap.parse_args()
    from dualip_amd.objectives.matching import MatchingInputArgs
    from dualip_amd.projections import create_projection_map

    dev = "cuda:0"
    t0 = time.perf_counter()
    A, C, counts = generate(device=dev)
    torch.cuda.synchronize()
    n, m, nnz = A.shape[1], A.shape[0], A.values().numel()
    print(f"generated {n} users x {m} movies, {nnz} ratings in {time.perf_counter() - t0:.2f}s; ratings per user: min {int(counts.min())} "
          f"median {int(counts.median())} mean {float(counts.float().mean()):.0f} max {int(counts.max())}; "
          f"{float((counts > 253).float().mean()) * 100:.1f}% of the users (holding {float(counts[counts > 253].sum()) / nnz * 100:.0f}% of the ratings) exceed one 256-window")
    inp = MatchingInputArgs(A=A, c=C, projection_map=create_projection_map("simplex", {"z": 1.0}, n, indices=range(n)),
                            b_vec=torch.full((m,), args.capacity, device=dev), equality_mask=None)
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent

    t0 = time.perf_counter()
    f = MatchingSolverDualObjectiveFunction(matching_input_args=inp, gamma=args.gamma)
    torch.cuda.synchronize()
    print(f"objective built in {time.perf_counter() - t0:.3f}s: {f.info()}")
    solver = AcceleratedGradientDescent(max_iter=args.max_iter, gamma=args.gamma, initial_step_size=1e-8, max_step_size=1e-6, iteration_callback=False)
    t0 = time.perf_counter()
    res = solver.maximize(f, stress_duals(m, dev) if args.stress else torch.zeros(m, device=dev))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if os.environ.get("DUALIP_HIP_TIMELINE"):
        import numpy as np

        tl = f.timeline().astype(np.int64)
        us = (tl - tl[:, 0].min()) / 100.0
        d = us[:, 2] - us[:, 1]
        print(f"per-workgroup tile-loop duration of the last launch (us): min {d.min():.0f} mean {d.mean():.0f} max {d.max():.0f}; kernel span {us[:, 3].max():.0f}")
    if not args.no_verify:
        verify(f, inp, args.gamma, res.dual_val, dev)
    if args.measure_traffic:
        import json

        import bench

        child = [sys.executable, os.path.abspath(__file__), "--max-iter", "20", "--no-verify"] + (["--stress"] if args.stress else [])
        hbm, details = bench.measure_traffic_now(child, ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY"))
        info = f.info()
        streamed = (nnz + info["slice_elements"] - info["slice_nnz"]) * (8 + info["row_index_bytes"])
        rec = {"command": " ".join(child), "commit": bench.current_commit(), "hbm_bytes_per_launch": hbm, "counters": details, "layout": info,
               "values_and_rows_by_construction": streamed, "note": "hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950 corrections, "
               "profiles/r02_fetch_size_calibration.json); the working set (~125 MB) fits the 256 MB Infinity Cache, so HBM itself is not the bound"}
        json.dump(rec, open(args.measure_traffic, "w"), indent=1)
        print("measured:", json.dumps(rec)[:600])
    print(f"maximize: {args.max_iter} iterations in {dt:.3f}s ({args.max_iter / dt:.1f} iterations/s, {dt / args.max_iter * 1e3:.3f} ms each); dual objective "
          f"{res.dual_objective:.3f} (first {res.dual_objective_log[0]:.3f})")


if __name__ == "__main__":
    main()
