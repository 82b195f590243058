#!/usr/bin/env python3
"""Matching-LP benchmark driver: one script for the reference's two (benchmark/run_matching_benchmark.py:47-148 single GPU,
run_matching_benchmark_dist.py:33-190 one process per GPU under torchrun).

    python benchmark/run_matching_benchmark.py --num-sources 10000000 --max-iter 1000 --json-output out.json
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 benchmark/run_matching_benchmark.py --num-sources 100000000

Same parameters as the reference's benchmark/config.py (10 000 destinations, sparsity 1e-3, seed 42, float32, step sizes
1e-3 / 1e-1, gamma 1e-3, optional gamma continuation 35 / 0.7 ending at the final gamma, optional Jacobi preconditioning),
same flags (``--cache-dir --num-sources --max-iter --json-output``) and the same metrics keys in the JSON file.  What
differs is how the data gets to the GPUs: the reference builds the whole problem on rank 0 in Python, splits it and
scatters pickled shards (run_matching_benchmark_dist.py:43-110); here every rank generates exactly its own columns on
its device (benchmark/synthetic.py) -- or maps a reference-format disk cache (benchmark/cache_format.py) when
``--cache-dir`` holds one -- so no host-side copy of the 100M-entity problem ever exists.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

NUM_SOURCES = 25_000_000      # benchmark/config.py:9-22
NUM_DESTINATIONS = 10_000
TARGET_SPARSITY = 0.001
SEED = 42
DTYPE = torch.float32
MAX_ITER = 1000
INITIAL_STEP_SIZE = 1e-3
MAX_STEP_SIZE = 1e-1
FINAL_GAMMA = 1e-3            # run_matching_benchmark.py:26
GAMMA_DECAY_STEPS = 35
GAMMA_DECAY_FACTOR = 0.7


def initial_gamma(max_iter: int, use_decay: bool) -> float:
    """Start so that the continuation ends at FINAL_GAMMA (run_matching_benchmark.py:33-38)."""
    return FINAL_GAMMA / (GAMMA_DECAY_FACTOR ** (max_iter // GAMMA_DECAY_STEPS)) if use_decay else FINAL_GAMMA


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--cache-dir", type=str, default=None, help="directory of a reference-format memmap cache (single process only)")
    ap.add_argument("--num-sources", type=int, default=None)
    ap.add_argument("--max-iter", type=int, default=None)
    ap.add_argument("--json-output", type=str, default=None, help="save metrics to a JSON file (rank 0)")
    ap.add_argument("--gamma-decay", action="store_true", help="gamma continuation, 35 steps / factor 0.7 (USE_GAMMA_DECAY)")
    ap.add_argument("--precondition", action="store_true", help="Jacobi row normalisation (USE_PRECONDITIONING)")
    ap.add_argument("--projection", choices=["simplex", "box", "mixed"], default="simplex", help="reference default: one simplex z=1 over all sources")
    ap.add_argument("--fairness", type=float, default=None, metavar="DELTA",
                    help="add the two fairness rows of docs/demo/matching_complex.rst (first half of the sources against the second, tolerance DELTA); single GPU")
    args = ap.parse_args()
    n = args.num_sources or NUM_SOURCES
    max_iter = args.max_iter or MAX_ITER

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    import bench
    from benchmark import cache_format
    from benchmark.synthetic import CHUNK_COLS, generate_matching_problem
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction, MatchingSolverDualObjectiveFunctionDistributed
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent
    from dualip_amd.preprocessing.precondition import jacobi_precondition

    def log(msg):
        if rank == 0:
            print(msg, flush=True)

    log(f"sources={n} destinations={NUM_DESTINATIONS} sparsity={TARGET_SPARSITY} seed={SEED} gpus={world} max_iter={max_iter} "
        f"gamma={FINAL_GAMMA} decay={args.gamma_decay} precondition={args.precondition} projection={args.projection}")
    log("[1/3] Generating data...")
    t0 = time.perf_counter()
    ranges, pm_local = bench.shard_plan(args.projection, n, world, rank, CHUNK_COLS)
    inp = None
    if args.cache_dir and world == 1:
        inp = cache_format.load_matching_cache(args.cache_dir, n, NUM_DESTINATIONS, TARGET_SPARSITY, DTYPE, SEED, device=device, projection_map=pm_local)
        log("      loaded the disk cache" if inp is not None else "      no matching cache: generating")
    if inp is None:
        def reduce_loads(v):
            if world > 1:
                dist.all_reduce(v)
            return v

        inp = generate_matching_problem(n, NUM_DESTINATIONS, TARGET_SPARSITY, seed=SEED, device=device, dtype=DTYPE, col_ranges=ranges, reduce_loads=reduce_loads)["input_args"]
        inp.projection_map = pm_local
    if args.fairness is not None and world > 1:
        raise NotImplementedError("--fairness: wrap the per-rank objective as tests/test_gpu_two_ranks.py[fairness] does; the driver runs it on one GPU")
    if args.precondition:
        if world > 1:
            raise NotImplementedError("Jacobi row norms of a column-sharded matrix need one more all-reduce; run it single-GPU")
        jacobi_precondition(inp.A, inp.b_vec)  # in place (preprocessing/precondition.py:8-28)
    torch.cuda.synchronize()
    log(f"      {time.perf_counter() - t0:.3f}s")

    log("[2/3] Creating objective...")
    t0 = time.perf_counter()
    gamma0 = initial_gamma(max_iter, args.gamma_decay)
    if world > 1:
        b_vec, inp.b_vec = inp.b_vec, None
        objective = MatchingSolverDualObjectiveFunctionDistributed(inp, b_vec, gamma0, host_device=device)
    elif args.fairness is not None:
        from dualip_amd.objectives.matching_fairness import MatchingFairnessDualObjectiveFunction

        b_vec = torch.cat([inp.b_vec, torch.full((2,), float(args.fairness), dtype=DTYPE, device=device)])
        inp.b_vec = b_vec
        objective = MatchingFairnessDualObjectiveFunction(inp, gamma0, group_ratio=0.5)
        log(f"      fairness pair: {'streamed by the fused kernel' if objective.native else 'folded into the cost'}")
    else:
        b_vec = inp.b_vec
        objective = MatchingSolverDualObjectiveFunction(matching_input_args=inp, gamma=gamma0)
    torch.cuda.synchronize()
    log(f"      {time.perf_counter() - t0:.3f}s")

    log("[3/3] Running solver...")
    solver = AcceleratedGradientDescent(
        max_iter=max_iter, gamma=gamma0, initial_step_size=INITIAL_STEP_SIZE, max_step_size=MAX_STEP_SIZE,
        gamma_decay_type="step" if args.gamma_decay else None,
        gamma_decay_params={"decay_steps": GAMMA_DECAY_STEPS, "decay_factor": GAMMA_DECAY_FACTOR} if args.gamma_decay else None,
        save_primal=(world == 1), iteration_callback=False,
    )
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    result = solver.maximize(objective, torch.zeros_like(b_vec), rank=rank)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    solve_time = time.perf_counter() - t0

    if rank == 0:
        r = result.objective_result
        metrics = {
            "num_gpus": world,
            "num_sources": n,
            "num_destinations": NUM_DESTINATIONS,
            "target_sparsity": TARGET_SPARSITY,
            "max_iter": max_iter,
            "solve_time": solve_time,
            "dual_objective": float(result.dual_objective),
            "primal_objective": float(r.primal_objective) if r.primal_objective is not None else None,
            "reg_penalty": float(r.reg_penalty),
            "max_pos_slack": float(r.max_pos_slack),
            "sum_pos_slack": float(r.sum_pos_slack),
        }
        print(f"solve time {solve_time:.3f}s  ({max_iter / solve_time:.1f} iterations/s)  dual objective {metrics['dual_objective']:.6f}  "
              f"max_pos_slack {metrics['max_pos_slack']:.4g}  sum_pos_slack {metrics['sum_pos_slack']:.4g}")
        if args.json_output:
            with open(args.json_output, "w") as f:
                json.dump(metrics, f, indent=2)
            print(f"metrics saved to {args.json_output}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
