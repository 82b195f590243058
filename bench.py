#!/usr/bin/env python3
"""bench.py -- dual-ascent iterations/sec of the matching LP on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config.workload): the synthetic matching LP of the reference's benchmark (benchmark/config.py:9-22,
generate_synthetic_data.py) at BASELINE.json's headline size -- 100M entities x 10k destinations, sparsity 1e-3
(~1e9 non-zeros), mixed box / simplex projection map, gamma = 1e-3, fp32 -- column-sharded over the N GPUs
(strong scaling: the global problem is fixed).  One "step" = one full dual-ascent iteration: fused CSC pass
(gather, projection, scatter-add, reductions) + slab reduction + [RCCL sum-all-reduce of m+2 doubles when N > 1] +
device-side step-size/AGD update.  Inputs are resident in HBM before the timed region.

The JSON line carries, besides the contract fields:
  roofline     -- HBM roofline of the fused kernel: algorithmic bytes per launch (12 E + 4 n + 16 m) /
                  average launch duration measured with HIP events on the launch stream inside the timed region.
  cpu_baseline -- the CPU oracle (oracle/, OpenMP over all host cores) on a bounded sample of the same workload
                  (rank 0, N = 1 only), scaled to whole-problem iterations/s.  Reported baseline, not a target.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--entities", type=int, default=int(os.environ.get("DUALIP_BENCH_ENTITIES", 100_000_000)))
    ap.add_argument("--destinations", type=int, default=10_000)
    ap.add_argument("--sparsity", type=float, default=1e-3)
    ap.add_argument("--proj", choices=["mixed", "box", "simplex"], default="mixed")
    ap.add_argument("--gamma", type=float, default=1e-3)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-sample-cols", type=int, default=4_000_000)
    ap.add_argument("--cpu-sample-iters", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="take the N>1 code path (distributed objective + all-reduce) even with one rank")
    ap.add_argument("--emulate-world", type=int, default=0, help="developer aid: with --force-sharded and one rank, hold rank 0's shard of a W-rank run and "
                    "scale its partial sums by W in place of the all-reduce (per-rank cost of a W-GPU run; the printed value is NOT a result)")
    return ap.parse_args()


def projection_blocks(kind, n_global, align):
    """Blocks of the global projection map as (proj_type, params, lo, hi): config 4's "mixed" map is box[0,1] on the
    first half of the entities and simplex z=1 on the second (the cut sits on a generator-chunk boundary)."""
    if kind == "box":
        return [("box", {"lower": 0.0, "upper": 1.0}, 0, n_global)]
    if kind == "simplex":
        return [("simplex", {"z": 1.0}, 0, n_global)]
    half = (n_global // 2) // align * align if n_global >= 2 * align else n_global // 2
    return [("box", {"lower": 0.0, "upper": 1.0}, 0, half), ("simplex", {"z": 1.0}, half, n_global)]


def shard_plan(kind, n_global, world, rank, align):
    """(column ranges of this rank, local projection map): every rank takes its share of EVERY projection block
    (dualip_amd.utils.dist_utils.balanced_block_ranges), so all ranks carry the same operator mix."""
    from dualip_amd.projections import create_projection_map
    from dualip_amd.utils.dist_utils import balanced_block_ranges

    ranges, pm, pos = [], {}, 0
    for ptype, params, lo, hi in projection_blocks(kind, n_global, align):
        for a, b in balanced_block_ranges([(lo, hi)], world, rank, align):
            ranges.append((a, b))
            pm.update(create_projection_map(ptype, params, None, indices=range(pos, pos + (b - a))))
            pos += b - a
    return ranges, pm


def copy_ceiling_gbps(device, nbytes=1 << 30, reps=10):
    """Measured device-to-device copy rate (bytes read + bytes written per second) on this box: the practical HBM ceiling
    next to the 8 TB/s vendor peak (SURVEY.md 8d asks for both)."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def cpu_baseline(args, inp, pm_local, total_nnz):
    """Oracle (kind 'port') on a bounded sample of the local columns, all host cores.  The sample takes an equal share of
    columns from the head of EVERY projection entry (so a mixed map is sampled with its operator mix)."""
    import oracle
    from oracle import agd_oracle

    A = inp.A
    n_local = A.shape[1]
    entries = list(pm_local.items())
    per = max(1, min(args.cpu_sample_cols, n_local) // max(len(entries), 1))
    colptr_dev = A.ccol_indices()
    parts, projs, col_proj_parts = [], [], []
    for q, (_, e) in enumerate(entries):
        idx = e.indices
        lo_i = idx.start if isinstance(idx, range) else int(min(idx))
        hi_i = idx.stop if isinstance(idx, range) else int(max(idx)) + 1
        hi_i = min(hi_i, lo_i + per)
        cp = colptr_dev[lo_i : hi_i + 1].cpu().numpy().astype(np.int64)
        parts.append((cp, int(cp[0]), int(cp[-1])))
        projs.append((e.proj_type, e.proj_params))
        col_proj_parts.append(np.full(hi_i - lo_i, q, dtype=np.int32))
    ncols = int(sum(len(p[0]) - 1 for p in parts))
    colptr = np.zeros(ncols + 1, dtype=np.int64)
    pos, off = 0, 0
    rowidx_l, a_l, c_l = [], [], []
    for cp, k0, k1p in parts:
        cnt = len(cp) - 1
        colptr[pos + 1 : pos + cnt + 1] = cp[1:] - k0 + off
        pos += cnt
        off += k1p - k0
        rowidx_l.append(A.row_indices()[k0:k1p].cpu().numpy().astype(np.int64))
        a_l.append(A.values()[k0:k1p].cpu().numpy())
        c_l.append(inp.c.values()[k0:k1p].cpu().numpy())
    k1 = off
    rowidx, a, c = np.concatenate(rowidx_l), np.concatenate(a_l), np.concatenate(c_l)
    col_proj = np.concatenate(col_proj_parts)
    b = inp.b_vec.cpu().numpy()
    m = A.shape[0]
    npdt = np.float32 if args.dtype == "f32" else np.float64
    threads = oracle.max_threads()
    lam = np.zeros(m, dtype=npdt)
    sizer = agd_oracle.StepSizer(npdt)
    times = []
    for it in range(args.cpu_sample_iters + 1):
        t0 = time.perf_counter()
        ax, obj0, ssq, _ = oracle.matching_calculate(m, ncols, colptr, rowidx, a, c, lam, args.gamma, projs, col_proj=col_proj, dtype=npdt, want_x=False, threads=threads)
        grad, *_ = agd_oracle.epilogue(ax, obj0, ssq, lam, b, args.gamma, npdt)
        step = sizer(grad, lam, 1e-3, 1e-1)
        lam = np.maximum(lam + grad * npdt(step), 0).astype(npdt)
        times.append(time.perf_counter() - t0)
    per_iter = float(np.mean(times[1:]))
    sample_its = 1.0 / per_iter
    return {
        "value": sample_its * (k1 / max(total_nnz, 1)),
        "unit": "iterations/s",
        "cores": threads,
        "kind": "port",
        "sample": f"oracle/ (C, OpenMP {threads} threads) on {ncols} entities of the same problem ({k1} non-zeros; the first {per} of each of the "
        f"{len(entries)} projection blocks), {args.cpu_sample_iters} iterations at {per_iter * 1e3:.1f} ms = {args.cpu_sample_iters * per_iter * threads:.0f} core-seconds; "
        f"value = sample it/s x sample_nnz/total_nnz",
        "sample_ms_per_iteration": per_iter * 1e3,
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # developer aid: DUALIP_BENCH_ONE_DEVICE=1 runs every rank on cuda:0 with gloo collectives (RCCL refuses two ranks on one
    # device) -- a functional check of the N > 1 harness on a single-GPU box; the number it prints is not a result
    one_device = os.environ.get("DUALIP_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from benchmark.synthetic import CHUNK_COLS, generate_matching_problem
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction, MatchingSolverDualObjectiveFunctionDistributed
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent

    n, m = args.entities, args.destinations
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    # chunk-aligned column ranges: this rank's share of every projection block of the SAME global problem
    emu = args.emulate_world if (args.emulate_world > 1 and world == 1 and sharded) else 0
    ranges, pm_local = shard_plan(args.proj, n, emu or world, rank, CHUNK_COLS)

    def reduce_loads(v):
        if sharded:
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
        return v * float(emu) if emu else v

    t_gen = time.perf_counter()
    prob = generate_matching_problem(n, m, args.sparsity, seed=args.seed, device=device, dtype=tdt, col_ranges=ranges, reduce_loads=reduce_loads)
    inp = prob["input_args"]
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    inp.projection_map = pm_local
    nnz_local = prob["nnz"]
    nnz_t = torch.tensor([nnz_local], dtype=torch.float64, device=device)
    if sharded:
        dist.all_reduce(nnz_t)
    total_nnz = int(nnz_t.item())

    t_setup = time.perf_counter()
    b_vec = inp.b_vec
    if sharded:
        inp.b_vec = None
        f = MatchingSolverDualObjectiveFunctionDistributed(inp, b_vec, args.gamma, host_device=device)
        local = f.local_objective
        if emu:
            real_exchange = f._exchange
            f._exchange = lambda packed: real_exchange(packed).mul_(float(emu))
    else:
        f = MatchingSolverDualObjectiveFunction(inp, args.gamma)
        local = f
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    total_iters = args.warmup + args.steps
    solver = AcceleratedGradientDescent(
        max_iter=total_iters, gamma=args.gamma, initial_step_size=1e-3, max_step_size=1e-1, iteration_callback=False
    )
    run = solver.start_device_run(f, torch.zeros(m, dtype=tdt, device=device), rank=rank)
    run.advance(args.warmup)

    def fence():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier(**({} if one_device else {"device_ids": [local_rank]}))
        torch.cuda.synchronize()

    fence()
    local.profile(True)
    gc.disable()  # (the sharded loop issues every iteration from Python: keep collector pauses out of the timed region)
    t0 = time.perf_counter()
    run.advance(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    launches, kernel_ms = local.profile_read()
    local.profile(False)
    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if sharded:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    result = run.finish()
    run.close()

    vs = 4 if args.dtype == "f32" else 8
    alg_bytes = nnz_local * (2 * vs + 4) + prob["n_local"] * 4 + 4 * m * vs  # a, c, 32-bit row per nnz; colptr; lambda/grad/b/y
    avg_kernel_s = (kernel_ms / max(launches, 1)) * 1e-3
    achieved = alg_bytes / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"{args.proj}_{args.entities}_{world}")
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            "metric": "dual_ascent_iterations_per_sec",
            "value": args.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"synthetic matching LP (reference benchmark generator model), {n} entities x {m} destinations, sparsity {args.sparsity}, "
                f"{args.proj} projection map, gamma={args.gamma}, column-sharded over {world} GPU(s)",
                "entities": n,
                "destinations": m,
                "nnz": total_nnz,
                "projection": args.proj,
                "parallelism": f"column-shard x{world}",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "matching_fused_kernel",
                "kernel_avg_ms": avg_kernel_s * 1e3,
                "kernel_launches": launches,
                "algorithmic_bytes_per_launch": alg_bytes,
            },
            "aux": {
                "generate_s": t_gen,
                "setup_s": t_setup,
                "final_dual_objective": result.dual_objective,
                "layout": local.info(),
                "whole_iteration_GBps": alg_bytes * args.steps / elapsed / 1e9,
                "copy_ceiling_GBps": copy_ceiling_gbps(device),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            inp.b_vec = b_vec
            out["cpu_baseline"] = cpu_baseline(args, inp, pm_local, total_nnz)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if sharded:
        dist.barrier(**({} if one_device else {"device_ids": [local_rank]}))
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
