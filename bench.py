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

    python bench.py --gpus N            (WORLD_SIZE unset: re-executes itself under torch.distributed.run with N ranks)

The JSON line carries, besides the contract fields:
  roofline     -- HBM roofline of the fused kernel from the bytes the launch PHYSICALLY moves (values, 2-byte row indices,
                  descriptors, gradient slabs; counters when profiles/traffic.json has this configuration) / average launch
                  duration measured with HIP events on the launch stream inside the timed region.  The figure from SURVEY.md
                  8d's algorithmic bytes (12 E + 4 n + 16 m: 4-byte indices and column pointers the kernel does not read) is
                  aux.algorithmic_roofline; it can exceed 1 for that reason and is not a physical fraction.
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
    ap.add_argument("--cpu-sample-only", action="store_true", help="CPU legs on bounded samples only (no whole-problem runs even when the host has the memory)")
    ap.add_argument("--cpu-ref-cols", type=int, default=10_000_000, help="entities of the sample the reference-path CPU leg runs on")
    ap.add_argument("--cpu-ref-iters", type=int, default=3)
    ap.add_argument("--cpu-ref-threads", type=int, default=32, help="torch threads of the reference-path CPU leg (its measured optimum on the GPU box's host)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gamma-decay", action="store_true", help="BASELINE config 3: gamma continuation in the whole-solve leg (35 steps / factor 0.7, initial gamma = "
                    "gamma / 0.7^(iters // 35), benchmark/run_matching_benchmark.py:29-38); the timed window keeps the fixed gamma")
    ap.add_argument("--no-late", action="store_true", help="skip the whole-solve leg (the reference's 1000-iteration configuration: aux.whole_solve / aux.late)")
    ap.add_argument("--no-verify", action="store_true", help="skip the correctness leg at the benchmark size (aux.verified)")
    ap.add_argument("--solve-iters", type=int, default=1000, help="iterations of the whole-solve leg (benchmark/config.py:16-18: 1000)")
    ap.add_argument("--local-blocks", type=int, default=1, help="N > 1 route: split every rank's shard into this many kernel handles (RCCL: the collectives of all but "
                    "the last overlap the next block's fused pass)")
    ap.add_argument("--comm", choices=["auto", "p2p", "p2p-fenced", "rccl"], default=None, help="exchange back-end of the N > 1 route (default: DUALIP_COMM or auto)")
    ap.add_argument("--partition", choices=["contiguous", "reference", "balanced"], default="balanced", help="how the entities are split over the ranks (measured per-rank costs: profiles/r03_partitions_emulated_1gpu.md).  contiguous: "
                    "one contiguous column range per rank, as the reference (dist_utils.py:53-57), cut so that the ranks' COSTS are equal (a simplex column weighs "
                    "dist_utils.PROJECTION_COST of a box column); reference: the reference's count-balanced contiguous cuts n // W (+1); balanced: every rank takes "
                    "its share of every projection block (interleaved, not contiguous; run_solver: ComputeArgs(partition=\"balanced\")) -- the default: every rank "
                    "carries the same operator mix, which the fused pass streams fastest")
    ap.add_argument("--no-partition-compare", action="store_true", help="N > 1: skip the extra window that times the OTHER split (the reference's n // W (+1) cut when "
                    "--partition balanced, and vice versa) -- aux.partition.compared")
    ap.add_argument("--measure-traffic", action="store_true", help="measure roofline.traffic in THIS invocation: two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of a "
                    "3-step run of the same workload as child processes, 2 x FETCH_SIZE + WRITE_SIZE KiB per fused launch (gfx950 corrections of "
                    "MI355X_MICROARCH.md; calibration profiles/r02_fetch_size_calibration.json)")
    ap.add_argument("--no-traffic-fallback", action="store_true", help="do not run the counter passes when profiles/traffic.json has no record of this build (roofline.traffic is then the bytes by construction)")
    ap.add_argument("--record-traffic", action="store_true", help="with --measure-traffic: store the figure, the commit and the kernel layout it belongs to in profiles/traffic.json")
    ap.add_argument("--release-inputs", action="store_true", help="N = 1: after every other leg, make the kernel handle self-contained (objective.release_inputs()), drop the "
                    "generator's tensors, and time the window again -- aux.footprint: resident bytes before / after next to the bytes one launch streams")
    ap.add_argument("--emulate-rank", type=int, default=-1, help="with --emulate-world: which rank's shard to hold (default: the most expensive one of the partition)")
    ap.add_argument("--force-sharded", action="store_true", help="take the N>1 code path (distributed objective + exchange) even with one rank")
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


def partition_table(kind, n_global, world, align):
    """For every partition: the cut points (contiguous kinds) and the estimated per-rank cost in box-column units."""
    from dualip_amd.utils.dist_utils import contiguous_cuts, projection_cost, shard_costs

    blocks = projection_blocks(kind, n_global, align)
    cost_blocks = [(lo, hi, projection_cost(ptype)) for ptype, _, lo, hi in blocks]
    total = sum((hi - lo) * w for lo, hi, w in cost_blocks)
    out = {}
    for name in ("contiguous", "reference"):
        cuts = contiguous_cuts(n_global, world, cost_blocks if name == "contiguous" else ())
        costs = shard_costs(cuts, cost_blocks)
        out[name] = {"cuts": cuts, "costs": costs, "imbalance": max(costs) / (total / world)}
    out["balanced"] = {"cuts": None, "costs": [total / world] * world, "imbalance": 1.0}
    return out


def shard_plan(kind, n_global, world, rank, align, partition="contiguous"):
    """(column ranges of this rank, local projection map).

    contiguous / reference: ONE contiguous range [t_r, t_{r+1}) per rank (dualip_amd.utils.dist_utils.contiguous_cuts: cost-weighted
    or the reference's count-balanced cuts); the local map is the global one re-based.
    balanced: every rank takes its share of EVERY projection block (dist_utils.balanced_block_ranges)."""
    from dualip_amd.projections import create_projection_map
    from dualip_amd.utils.dist_utils import balanced_block_ranges

    ranges, pm, pos = [], {}, 0
    blocks = projection_blocks(kind, n_global, align)
    if partition == "balanced":
        for ptype, params, lo, hi in blocks:
            for a, b in balanced_block_ranges([(lo, hi)], world, rank, align):
                ranges.append((a, b))
                pm.update(create_projection_map(ptype, params, None, indices=range(pos, pos + (b - a))))
                pos += b - a
        return ranges, pm
    cuts = partition_table(kind, n_global, world, align)[partition]["cuts"]
    t0, t1 = cuts[rank], cuts[rank + 1]
    for ptype, params, lo, hi in blocks:
        a, b = max(lo, t0), min(hi, t1)
        if b > a:
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


def _mem_available_gb():
    try:
        with open("/proc/meminfo") as fh:
            for line in fh:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def _cpu_sample(args, inp, pm_local, n_cols):
    """The first n_cols / #entries columns of EVERY projection entry, on the host (so a mixed map is sampled with its mix); n_cols >= the
    shard's columns: every entry whole, i.e. the WHOLE problem."""
    A = inp.A
    n_local = A.shape[1]
    entries = list(pm_local.items())
    per = n_local if n_cols >= n_local else max(1, min(n_cols, n_local) // max(len(entries), 1))
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
        rowidx_l.append(np.asarray(A.row_indices()[k0:k1p].cpu().numpy(), dtype=np.int64))  # (no second copy when the indices are int64 already)
        a_l.append(A.values()[k0:k1p].cpu().numpy())
        c_l.append(inp.c.values()[k0:k1p].cpu().numpy())
    return dict(ncols=ncols, nnz=off, per=per, colptr=colptr, rowidx=np.concatenate(rowidx_l), a=np.concatenate(a_l), c=np.concatenate(c_l),
                col_proj=np.concatenate(col_proj_parts), projs=projs, b=inp.b_vec.cpu().numpy(), m=A.shape[0], n_entries=len(entries))


def read_ceiling_gbps(device, nbytes=10 << 30, reps=5):
    """Streaming READ rate of this box, best of five access shapes (16-byte non-temporal loads; one stream, or the fused kernel's three
    streams side by side with two, four or eight steps of a lane in flight, or with chunks CLAIMED dynamically by the workgroups -- the XCDs
    stream at different speeds, a static deal ends in the slow ones' tail -- dl_measure_read_bandwidth): a PROBE, not a proven ceiling.
    The default buffer is as large as the headline launch's stream (10 GiB; the 4 GiB, two-deep probe of round 4 read 5 % slower than the
    kernel it was held against).  `physical_frac` should be read against it besides the 8 TB/s of the data sheet."""
    import ctypes

    from dualip_amd import _hip

    free, _ = torch.cuda.mem_get_info(device)
    nbytes = int(min(nbytes, max(1 << 20, free // 2))) // 4096 * 4096
    buf = torch.empty(nbytes // 4, dtype=torch.float32, device=device).fill_(0.5)
    out = ctypes.c_double(0.0)
    with torch.cuda.device(device):
        _hip.check(_hip.load().dl_measure_read_bandwidth(_hip.ptr(buf), nbytes, reps, ctypes.byref(out), _hip.stream_ptr(device)))
    return float(out.value)


def cpu_baseline(args, inp, pm_local, total_nnz):
    """Two CPU legs on bounded samples of the same problem, all host cores, outside every timed region (reported baseline only):
      value   -- the reference's OP SEQUENCE restated in torch-on-CPU (oracle/torch_path.py: padded dense blocks per nnz bucket,
                 sort + cumsum simplex -- what device="cpu" executes in the reference; pinned to its goldens), >= 10M entities;
      c_port  -- the C oracle (oracle/matching_oracle.c, OpenMP), a per-column loop: much faster than the reference's path.
    Each leg runs on the WHOLE problem when the host has the memory (`extrapolated: false`), else on a sample scaled by nnz and labelled extrapolated."""
    import oracle
    from oracle import agd_oracle

    npdt = np.float32 if args.dtype == "f32" else np.float64
    threads = oracle.max_threads()
    host_cores = os.cpu_count() or 0  # (hardware threads of the host; `cores` below = the threads each leg actually used)
    out = {"unit": "iterations/s", "cores": threads, "host_cores": host_cores, "kind": "port", "extrapolated": True}
    # ---- C port -------------------------------------------------------------------------------------------------
    # The WHOLE problem when the host has room for it (one device-to-host copy of the CSC arrays, 17 GB at the headline; BASELINE.md section 3:
    # "100 M -- 2 iterations if host RAM allows"): measured, not extrapolated.  Otherwise the bounded sample, scaled by nnz and labelled so.
    host_bytes = inp.A.values().numel() * (8 + 2 * inp.A.values().element_size()) + inp.A.shape[1] * 8
    whole = not args.cpu_sample_only and _mem_available_gb() * 1e9 > 4.0 * host_bytes + 20e9
    smp = _cpu_sample(args, inp, pm_local, inp.A.shape[1] if whole else args.cpu_sample_cols)
    m = smp["m"]
    lam = np.zeros(m, dtype=npdt)
    sizer = agd_oracle.StepSizer(npdt)
    times = []
    c_iters = 2 if whole else args.cpu_sample_iters
    for it in range(c_iters + 1):
        t0 = time.perf_counter()
        ax, obj0, ssq, _ = oracle.matching_calculate(m, smp["ncols"], smp["colptr"], smp["rowidx"], smp["a"], smp["c"], lam, args.gamma, smp["projs"], col_proj=smp["col_proj"],
                                                     dtype=npdt, want_x=False, threads=threads)
        grad, *_ = agd_oracle.epilogue(ax, obj0, ssq, lam, smp["b"], args.gamma, npdt)
        step = sizer(grad, lam, 1e-3, 1e-1)
        lam = np.maximum(lam + grad * npdt(step), 0).astype(npdt)
        times.append(time.perf_counter() - t0)
    per_iter = float(np.mean(times[1:]))
    is_whole = smp["nnz"] == total_nnz
    out["c_port"] = {
        "value": (1.0 / per_iter) * (smp["nnz"] / max(total_nnz, 1)),
        "extrapolated": not is_whole,
        "sample": f"oracle/matching_oracle.c (OpenMP {threads} threads) on " + ("the WHOLE problem: " if is_whole else "") + f"{smp['ncols']} entities ({smp['nnz']} non-zeros"
        + ("" if is_whole else f"; the first {smp['per']} of each of the {smp['n_entries']} projection blocks") + f"), {c_iters} iterations after one warm-up at {per_iter * 1e3:.1f} ms "
        f"= {c_iters * per_iter * threads:.0f} core-seconds" + ("; measured, not extrapolated" if is_whole else "; value = sample it/s x sample_nnz / total_nnz"),
        "sample_ms_per_iteration": per_iter * 1e3,
    }
    # ---- the reference's op sequence (torch on CPU) --------------------------------------------------------------------
    import torch as _t

    from oracle.torch_path import ReferencePathObjective

    if smp["ncols"] != min(args.cpu_ref_cols, inp.A.shape[1]):
        whole_smp = smp if smp["nnz"] == total_nnz else None  # (kept for the whole-problem attempt below)
        smp = _cpu_sample(args, inp, pm_local, args.cpu_ref_cols)
    else:
        whole_smp = None
    # thread count: measured on the 256-thread host of the GPU box (tools/cpu_path_threads.py, 2M entities): 8 threads 469 ms,
    # 32 -> 373 ms, 64 -> 749 ms, 128 -> 1592 ms, 256 -> 41 s per iteration -- the op sequence is made of many small
    # memory-bound tensor ops that stop scaling early.  The leg runs at its best setting and says so in `cores`.
    old_threads = _t.get_num_threads()
    ref_threads = max(1, min(args.cpu_ref_threads, os.cpu_count() or 1))
    _t.set_num_threads(ref_threads)
    try:
        bounds = np.cumsum([0] + [int((smp["col_proj"] == q).sum()) for q in range(smp["n_entries"])])
        entries = [(pt, pp, np.arange(bounds[q], bounds[q + 1])) for q, (pt, pp) in enumerate(smp["projs"])]
        ref = ReferencePathObjective(m, smp["ncols"], smp["colptr"], smp["rowidx"], smp["a"], smp["c"], entries, args.gamma, dtype=_t.float32 if args.dtype == "f32" else _t.float64)
        lam_t = _t.zeros(m, dtype=_t.float32 if args.dtype == "f32" else _t.float64)
        b_t = _t.as_tensor(smp["b"]).to(lam_t.dtype)
        times = []
        for it in range(args.cpu_ref_iters + 1):
            t0 = time.perf_counter()
            ax, obj0, ssq, _ = ref.calculate(lam_t)
            lam_t = (lam_t + (ax - b_t) * 1e-3).clamp(min=0)  # (a plain projected ascent step: the m-sized side is negligible here)
            times.append(time.perf_counter() - t0)
        per_ref = float(np.mean(times[1:]))
        out["value"] = (1.0 / per_ref) * (smp["nnz"] / max(total_nnz, 1))
        out["sample"] = (f"oracle/torch_path.py -- the reference's calculate() op sequence in torch on CPU, {_t.get_num_threads()} threads -- on {smp['ncols']} entities "
                         f"({smp['nnz']} non-zeros; the first {smp['per']} of each of the {smp['n_entries']} projection blocks), {args.cpu_ref_iters} iterations at {per_ref * 1e3:.0f} ms; "
                         f"value = sample it/s x sample_nnz / total_nnz (extrapolated to the whole problem)")
        out["sample_ms_per_iteration"] = per_ref * 1e3
        out["cores"] = ref_threads
        out["c_port"]["cores"] = threads
        # ---- the same op sequence on the WHOLE problem, ONE iteration, when the host has the memory (MemAvailable >= 150 GB) and the
        #      sample predicts it fits a minute: then `value` is that measurement and nothing is extrapolated ----
        predicted = per_ref * total_nnz / max(smp["nnz"], 1)
        out["extrapolated"] = smp["nnz"] != total_nnz
        if not out["extrapolated"]:
            pass  # (the sample WAS the whole problem: nothing scaled)
        elif whole_smp is not None and _mem_available_gb() >= 150.0 and predicted <= 60.0:
            try:
                del ref
                w = whole_smp
                bounds = np.cumsum([0] + [int((w["col_proj"] == q).sum()) for q in range(w["n_entries"])])
                entries = [(pt, pp, np.arange(bounds[q], bounds[q + 1])) for q, (pt, pp) in enumerate(w["projs"])]
                t0 = time.perf_counter()
                ref = ReferencePathObjective(m, w["ncols"], w["colptr"], w["rowidx"], w["a"], w["c"], entries, args.gamma, dtype=lam_t.dtype)
                t_setup = time.perf_counter() - t0
                lam_w = _t.zeros(m, dtype=lam_t.dtype)
                t0 = time.perf_counter()
                ax, obj0, ssq, _ = ref.calculate(lam_w)
                lam_w = (lam_w + (ax - b_t) * 1e-3).clamp(min=0)
                t_one = time.perf_counter() - t0
                out["sample_extrapolated_value"] = out["value"]
                out["value"] = 1.0 / t_one
                out["extrapolated"] = False
                out["sample"] = (f"oracle/torch_path.py -- the reference's calculate() op sequence in torch on CPU, {_t.get_num_threads()} threads -- on the WHOLE problem "
                                 f"({w['ncols']} entities, {w['nnz']} non-zeros): ONE iteration (the first, allocations included) at {t_one:.1f} s after {t_setup:.1f} s of bucket set-up; "
                                 f"measured, not extrapolated.  The 10M-entity sample ({args.cpu_ref_iters} iterations at {per_ref * 1e3:.0f} ms) scaled by nnz predicts "
                                 f"{out['sample_extrapolated_value']:.4f} it/s")
                out["sample_ms_per_iteration"] = t_one * 1e3
            except Exception as exc:  # (memory: keep the sample's figure, say why)
                out["whole_problem_attempt"] = f"failed: {type(exc).__name__}: {str(exc)[:200]}"
        elif whole_smp is None:
            out["whole_problem_attempt"] = "not attempted: the C leg ran on a sample (host memory)"
        else:
            out["whole_problem_attempt"] = f"not attempted: MemAvailable {_mem_available_gb():.0f} GB, predicted {predicted:.0f} s per iteration"
    finally:
        _t.set_num_threads(old_threads)
    return out


TRAFFIC_LAYOUT_KEYS = ("tiles", "long_columns", "workgroup_columns", "slices", "slice_elements", "slice_lane_columns", "window_descriptor_words", "row_index_bytes", "hot_rows", "layout", "workgroups",
                       "slab_bytes")


def library_source_hash():
    """Digest of the sources libdualip_hip.so was built from (dualip_amd/_build.py: every .hip / .h and the flags).  A traffic record belongs
    to ONE such digest: any kernel edit, whether or not it changes the layout dl_matching_info reports, retires the record."""
    from dualip_amd import _build

    try:
        with open(_build.HASH_PATH) as fh:
            return fh.read().strip()
    except OSError:
        return _build.source_hash()


def traffic_key(args, world):
    key = f"{args.proj}_{args.entities}_{world}"
    if args.sparsity != 1e-3 or args.destinations != 10_000:
        key += f"_m{args.destinations}_s{args.sparsity:g}"
    return key + ("_f64" if args.dtype == "f64" else "")


def current_commit():
    c = os.environ.get("DUALIP_COMMIT")
    if c:
        return c
    try:
        import subprocess

        return subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True, timeout=10).stdout.strip() or "unknown"
    except Exception:
        return "unknown"


def stored_traffic(args, world, lay, phys_bytes):
    """(bytes, source) from profiles/traffic.json, or (None, why not).  A record carries the SOURCE HASH of the kernels it was measured on
    (any kernel edit retires it, whether or not the layout moves) and the kernel layout (dl_matching_info); it is REFUSED when either differs
    from this run's.  Bare numbers of rounds 1-3 are refused too."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(tpath)).get(traffic_key(args, world))
    except Exception:
        rec = None
    if rec is None:
        return None, "no counter pass recorded for this configuration"
    if isinstance(rec, dict):
        have = library_source_hash()
        if rec.get("srchash") != have:
            return None, (f"the record of commit {rec.get('commit')} was taken on another build of the kernels (source hash {str(rec.get('srchash'))[:12]} against {have[:12]} "
                          "of this library): refused")
        diff = {k: (rec.get("layout", {}).get(k), lay.get(k)) for k in TRAFFIC_LAYOUT_KEYS if rec.get("layout", {}).get(k) != lay.get(k)}
        if diff:
            return None, f"the record of commit {rec.get('commit')} was taken on another kernel layout ({diff}): refused"
        return float(rec["bytes"]), (f"profiles/traffic.json: rocprofv3 --pmc passes of this command at commit {rec.get('commit')} ({rec.get('file', 'bench.py --measure-traffic --record-traffic')}), "
                                     f"2 x FETCH_SIZE + WRITE_SIZE (gfx950 corrections), same kernel sources (hash {have[:12]}) and layout as this run")
    return None, "the record is a bare number of rounds 1-3 (no kernel source hash, no layout): refused"


def measure_traffic_now(child=None, extra_counters=()):
    """HBM bytes per fused launch measured NOW: this command line re-run (3 steps, no side legs) under `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (separate passes, as the guide prescribes), averaged over the fused kernel's dispatches.  Returns (bytes, details).
    `child`: another command to profile instead (benchmark/movielens_like.py passes its own); `extra_counters`: more single-counter passes
    whose per-launch means are added to the details."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    drop = {"--measure-traffic", "--record-traffic", "--no-traffic-fallback"}
    base = [a for a in sys.argv[1:] if a not in drop]
    for flag in ("--steps", "--warmup"):  # (value flags the child gets its own of)
        while flag in base:
            i = base.index(flag)
            del base[i:i + 2]
    if child is None:
        child = [sys.executable, os.path.abspath(__file__)] + base + ["--steps", "3", "--warmup", "1", "--no-late", "--no-verify", "--no-cpu-baseline", "--no-traffic-fallback"]
    vals, details = {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE") + tuple(extra_counters):
        tmp = tempfile.mkdtemp(prefix="dualip_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp", DUALIP_BENCH_NO_EVENTS="1")
            try:
                r = subprocess.run([rocprof, "--output-format", "csv", "--pmc", counter, "-d", tmp, "-o", "p", "--"] + child, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            except subprocess.TimeoutExpired:
                if counter in ("FETCH_SIZE", "WRITE_SIZE"):
                    return None, {"error": f"the {counter} pass did not finish in 600 s"}
                details[counter + "_error"] = "pass timed out"
                continue
            rows = []
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "matching_fused" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        rows.append(float(row["Counter_Value"]))
            if not rows:
                if counter in ("FETCH_SIZE", "WRITE_SIZE"):
                    return None, {"error": f"no {counter} rows for the fused kernel (rocprofv3 exit {r.returncode}): {r.stderr[-300:]}"}
                details[counter + "_error"] = f"no rows (rocprofv3 exit {r.returncode})"
                continue
            vals[counter] = sum(rows) / len(rows)
            details[counter + ("_KiB_per_launch" if counter.endswith("_SIZE") else "_per_launch")] = vals[counter]
            details[counter + "_dispatches"] = len(rows)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, details


def timed_window(run, local, comm, n_iters, fence, elapsed_max, stride=1):
    """Time `n_iters` iterations of a device run between two fences; returns (seconds [max over ranks], fused launches,
    fused-kernel ms, exchange brackets, exchange ms)."""
    fence()
    # every pair of event records costs ~5 us of stream time (measured: 9.5 us per iteration with two pairs, 0.6 % of a 100M
    # iteration and 4 % of a 12.5M one): the hooks bracket every `stride`-th launch
    events = 0 if os.environ.get("DUALIP_BENCH_NO_EVENTS") == "1" else stride
    local.profile(events)
    if comm is not None:
        comm.profile(events)
    gc.disable()
    t0 = time.perf_counter()
    run.advance(n_iters)
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    launches, kernel_ms = local.profile_read()
    local.profile(False)
    xn, xms = (0, 0.0)
    if comm is not None:
        xn, xms = comm.profile_read()
        comm.profile(False)
    return elapsed_max(elapsed), launches, kernel_ms, xn, xms


def _free_port():
    import socket

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def respawn_under_torchrun(args):
    """``python bench.py --gpus N`` with no launcher around it: run the same command line as N ranks under
    torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1) and hand its output through -- rank 0 prints the
    ONE JSON line.  Returns the launcher's exit status."""
    import subprocess

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("MASTER_PORT", None)
    if os.environ.get("DUALIP_BENCH_ONE_DEVICE") == "1" and args.gpus >= 8:
        # developer mode, eight or more ranks on ONE device: a ninth process with a copy-engine queue oversubscribes the driver's run list and the
        # exchange's in-kernel waits stall (profiles/r05_world8_on_one_gpu.md); the ranks copy with blit kernels instead.  One rank per GPU: untouched.
        env.setdefault("HSA_ENABLE_SDMA", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def rank_identities(device, rank, world, one_device):
    """Which PHYSICAL GPU every rank runs on, gathered over the side channel before anything is timed: UUID, PCI address, ordinal, host, pid
    (benchmark/run_matching_benchmark_dist.py:35-41 pins rank r to cuda:r and trusts it; a scaling line should prove it).  Raises unless the
    ranks sit on `world` distinct devices -- except in the developer mode that shares one on purpose (DUALIP_BENCH_ONE_DEVICE=1)."""
    import socket

    props = torch.cuda.get_device_properties(device)
    pci = None
    if all(hasattr(props, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        pci = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}"
    mine = {"rank": rank, "host": socket.gethostname(), "pid": os.getpid(), "device_ordinal": device.index, "name": props.name,
            "uuid": str(getattr(props, "uuid", None)), "pci": pci, "visible": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    keys = [(e["host"], e["uuid"] if e["uuid"] not in (None, "None") else (e["pci"] or e["device_ordinal"])) for e in everyone]
    distinct = len(set(keys))
    if distinct != world and not one_device:
        raise RuntimeError(f"bench.py --gpus {world}: the ranks run on {distinct} distinct GPU(s), not {world} ({everyone}); set DUALIP_BENCH_ONE_DEVICE=1 "
                           "for the developer mode that shares a device on purpose")
    return {"ranks": everyone, "distinct_gpus": distinct, "one_device_mode": bool(one_device),
            "process_group": {"backend": dist.get_backend(), "world_size": dist.get_world_size()}}


def collective_selftest(comm, m, device, rank, world):
    """Before anything is timed: this library's exchange (dl_allreduce_sum on the communicator the solve will use) against
    torch.distributed's all_reduce of the same random vector, on every rank.  Returns a dict for aux.collective."""
    out = {"backend": comm.backend if comm is not None else "torch.distributed", "selftest": None}
    if comm is None:
        return out
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    worst = 0.0
    for rnd in range(4):  # (both mailbox parities, twice)
        v = torch.randn(m + 2, dtype=torch.float64, generator=g).to(device)
        ref = v.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        ours = comm.all_reduce_(v.clone())
        comm.check()
        worst = max(worst, float((ours - ref).abs().max() / ref.abs().max().clamp_min(1e-300)))
    agree = torch.tensor([worst], dtype=torch.float64, device=device)
    dist.all_reduce(agree, op=dist.ReduceOp.MAX)
    worst = float(agree.item())
    out["selftest"] = {"against": "torch.distributed.all_reduce", "rounds": 4, "max_rel_err_any_rank": worst, "ok": bool(worst <= 1e-12)}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # developer aid: DUALIP_BENCH_ONE_DEVICE=1 runs every rank on cuda:0 with a gloo side channel (RCCL refuses two ranks on one
    # device; the P2P exchange does not) -- a functional check of the N > 1 harness on a single-GPU box; the number is not a result.
    # (Two ranks work; with four at 10M entities each a rank's kernel spins in its bounded wait while the GPU time-slices the other
    # processes' kernels, and the wait runs into its limit: an artefact of sharing one device, reported as such by dl_comm_check.)
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
    nb = max(1, args.local_blocks) if sharded else 1
    vworld = (emu or world) * nb  # a split shard = nb consecutive virtual ranks
    ptable = partition_table(args.proj, n, vworld, CHUNK_COLS)
    emu_rank = 0
    if emu:  # hold the most expensive shard of the partition unless told otherwise
        costs = ptable[args.partition]["costs"]
        emu_rank = args.emulate_rank if 0 <= args.emulate_rank < emu else max(range(emu), key=lambda r: sum(costs[r * nb:(r + 1) * nb]))
    vrank0 = (emu_rank if emu else rank) * nb

    def reduce_loads(v):
        if sharded:
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
        return v * float(emu) if emu else v

    def make_shard(partition):
        """This rank's columns under `partition` as nb kernel-handle inputs, with the capacity vector of the WHOLE problem."""
        block_inputs, nnz_local, loads, rho, ranges_first = [], 0, None, None, None
        for k in range(nb):  # (nb > 1: the shard as nb kernel handles, each with its share of every projection block)
            ranges_k, pm_k = shard_plan(args.proj, n, vworld, vrank0 + k, CHUNK_COLS, partition)
            ranges_first = ranges_k if k == 0 else ranges_first
            prob_k = generate_matching_problem(n, m, args.sparsity, seed=args.seed, device=device, dtype=tdt, col_ranges=ranges_k)
            prob_k["input_args"].projection_map = pm_k
            block_inputs.append(prob_k["input_args"])
            nnz_local += prob_k["nnz"]
            loads = prob_k["loads_local"] if loads is None else loads + prob_k["loads_local"]
            rho = prob_k["rho"]
        # b = rho * (greedy load of the WHOLE problem + 1e-8): the m-sized loads are the only thing the shards of the generator share
        b_vec = (torch.from_numpy(rho).to(device) * (reduce_loads(loads) + 1e-8)).to(tdt)
        for bi in block_inputs:
            bi.b_vec = b_vec
        return block_inputs, nnz_local, b_vec, ranges_first

    t_gen = time.perf_counter()
    block_inputs, nnz_local, b_vec, ranges_first = make_shard(args.partition)
    pm_local = block_inputs[0].projection_map
    inp = block_inputs[0]
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    nnz_t = torch.tensor([nnz_local], dtype=torch.float64, device=device)
    if sharded:
        dist.all_reduce(nnz_t)
    total_nnz = int(nnz_t.item())

    t_setup = time.perf_counter()
    comm, collective = None, None
    identities = rank_identities(device, rank, world, one_device) if sharded else None
    if sharded:
        for bi in block_inputs:
            bi.b_vec = None
        f = MatchingSolverDualObjectiveFunctionDistributed(block_inputs if nb > 1 else block_inputs[0], b_vec, args.gamma, host_device=device, comm_backend=args.comm)
        local = f.local_objective
        comm = f.communicator()  # (None: no native exchange here -- torch.distributed from Python, aux.collective says why)
        try:  # before anything is timed (and before the emulation factor is set)
            collective = collective_selftest(comm, m, device, rank, world)
        except Exception as exc:  # must show in the line, not kill the measurement
            collective = {"backend": comm.backend if comm is not None else "torch.distributed", "selftest": {"ok": False, "error": f"{type(exc).__name__}: {exc}"}}
        if emu and comm is not None:
            comm.set_emulation(float(emu))
    else:
        f = MatchingSolverDualObjectiveFunction(inp, args.gamma)
        local = f
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    def fence():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier(**({} if one_device else {"device_ids": [local_rank]}))
        torch.cuda.synchronize()

    def elapsed_max(sec):
        el = torch.tensor([sec], dtype=torch.float64, device=device)
        if sharded:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    vs = 4 if args.dtype == "f32" else 8
    nnz_first = int(block_inputs[0].A.values().numel())  # (the event hook brackets the FIRST block's fused launches)
    alg_bytes = nnz_first * (2 * vs + 4) + block_inputs[0].A.shape[1] * 4 + 4 * m * vs  # a, c, 32-bit row per nnz; colptr; lambda/grad/b/y
    lay = local.info()
    # what the launch physically moves through HBM: the two value arrays, the re-encoded row indices, the 48-byte window
    # descriptors, and the per-workgroup gradient slabs it writes (lambda and the projection table are L2-served re-reads)
    # -- plus, for columns held in column-per-lane slices, the padding of their transposed copy, 16 bytes per slice, one length
    # byte per column
    per_nnz = 2 * vs + lay["row_index_bytes"]
    desc_bytes = 4 * lay.get("window_descriptor_words", 12 if lay["layout"] == 4 else 4)
    phys_bytes = (nnz_first + lay.get("slice_elements", 0) - lay.get("slice_nnz", 0)) * per_nnz + (lay["tiles"] - lay["long_columns"]) * desc_bytes \
        + lay["long_columns"] * (48 if lay["layout"] == 4 else 16) + lay.get("slices", 0) * 16 + lay.get("slice_mixed_columns", 0) + lay["workgroups"] * (m * lay.get("slab_bytes", 8) + 16)

    # roofline.traffic: HBM bytes per launch from the PMC counters -- measured in this invocation (--measure-traffic), else the record
    # of profiles/traffic.json IF it was taken on this very kernel layout, else what the launch moves by construction (phys_bytes)
    traffic, traffic_source, traffic_details = None, None, None
    if args.measure_traffic and not sharded and rank == 0:
        traffic, traffic_details = measure_traffic_now()
        if traffic:
            traffic_source = f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (3 steps) at commit {current_commit()}, 2 x FETCH_SIZE + WRITE_SIZE KiB per launch"
            if args.record_traffic:
                tpath = os.path.join(ROOT, "profiles", "traffic.json")
                try:
                    table = json.load(open(tpath))
                except Exception:
                    table = {}
                table[traffic_key(args, world)] = {"bytes": traffic, "commit": current_commit(), "srchash": library_source_hash(), "layout": {k: lay.get(k) for k in TRAFFIC_LAYOUT_KEYS},
                                                   "by_construction": phys_bytes, "counters": traffic_details, "file": "bench.py --measure-traffic --record-traffic"}
                json.dump(table, open(tpath, "w"), indent=1)
    if not traffic and not sharded:  # (the recorded passes are single-GPU runs of the whole problem)
        traffic, traffic_source = stored_traffic(args, world, lay, phys_bytes)
        headline = args.entities == 100_000_000 and args.destinations == 10_000 and args.sparsity == 1e-3 and args.proj == "mixed" and args.dtype == "f32"
        if not traffic and rank == 0 and headline and not args.no_traffic_fallback and os.environ.get("DUALIP_BENCH_NO_EVENTS") != "1":
            # the driver's configuration and no record of THIS build: measure now (two counter passes of three steps of this command) rather than
            # quote bytes by construction (other sizes -- tests, tools -- keep the by-construction figure, labelled as such)
            refused = traffic_source
            traffic, traffic_details = measure_traffic_now()
            if traffic:
                traffic_source = (f"measured in this run (fallback: {refused}): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (3 steps), kernel source hash "
                                  f"{library_source_hash()[:12]}, 2 x FETCH_SIZE + WRITE_SIZE KiB per launch")
            else:
                traffic_source = f"{refused}; the fallback counter passes failed ({(traffic_details or {}).get('error')})"
    if not traffic:
        why = traffic_source or "sharded run"
        traffic = float(phys_bytes)
        traffic_source = f"layout: bytes the launch moves by construction ({why})"
    roof_bytes = float(traffic)

    def roof(kernel_ms, launches):
        """(average launch seconds, physical GB/s, algorithmic GB/s)"""
        avg_s = (kernel_ms / max(launches, 1)) * 1e-3
        if avg_s <= 0:
            return avg_s, 0.0, 0.0
        return avg_s, roof_bytes / avg_s / 1e9, alg_bytes / avg_s / 1e9

    # ---- headline: W untimed iterations from zero duals, then exactly K timed ----------------------------------
    total_iters = args.warmup + args.steps
    stride = 4 if nnz_first > 400_000_000 else 8  # (bracket every 4th / 8th fused launch: see timed_window)

    def headline(f_, local_, comm_):
        solver = AcceleratedGradientDescent(max_iter=total_iters, gamma=args.gamma, initial_step_size=1e-3, max_step_size=1e-1, iteration_callback=False)
        run = solver.start_device_run(f_, torch.zeros(m, dtype=tdt, device=device), rank=rank)
        try:
            run.advance(args.warmup)
            el, ln, kms, xn_, xms_ = timed_window(run, local_, comm_, args.steps, fence, elapsed_max, stride)
            return el, ln, kms, xn_, xms_, run.finish()  # (finish() of a sharded run meets the ranks: a failed exchange raises on every rank)
        finally:
            run.close()

    from dualip_amd.utils.comm import ExchangeError

    try:
        elapsed, launches, kernel_ms, xn, xms, result = headline(f, local, comm)
    except ExchangeError as exc:
        # first contact with real xGMI may show that the unfenced P2P ordering does not hold there: every exchange is checksummed, the
        # ranks stop together, an `auto` communicator moves to the fenced ordering and the measurement is REPEATED -- the line says so
        if comm is None or not comm.degrade():
            raise
        if collective is not None:
            collective["repeated_after"] = str(exc)
        elapsed, launches, kernel_ms, xn, xms, result = headline(f, local, comm)
    avg_kernel_s, achieved, achieved_alg = roof(kernel_ms, launches)
    per_rank = None
    if sharded:  # every rank's own fused-kernel average and exchange bracket in the timed window (the headline is the max over ranks of the wall clock)
        mine = torch.tensor([avg_kernel_s * 1e3, (xms / xn * 1e3) if xn else -1.0, float(nnz_local)], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = torch.stack(allr).cpu()
        kms_r, xus_r = rows[:, 0].tolist(), rows[:, 1].tolist()
        per_rank = {"kernel_avg_ms": kms_r, "kernel_avg_ms_min": min(kms_r), "kernel_avg_ms_max": max(kms_r), "kernel_skew": (max(kms_r) / min(kms_r) - 1.0) if min(kms_r) > 0 else None,
                    "us_per_exchange": [v if v >= 0 else None for v in xus_r], "nnz": [int(v) for v in rows[:, 2].tolist()]}

    # ---- the reference's whole solve (benchmark/config.py:16-18: max_iter 1000) and its late window ----------------
    late, whole, lam_late = None, None, result.dual_val
    if not args.no_late and args.solve_iters >= 200:
        S = args.solve_iters
        w0, w1 = int(S * 0.8), int(S * 0.9)
        decay_kw = {}
        gamma0 = args.gamma
        if args.gamma_decay:
            gamma0 = args.gamma / (0.7 ** (S // 35))
            decay_kw = dict(gamma_decay_type="step", gamma_decay_params={"decay_steps": 35, "decay_factor": 0.7})
        solver2 = AcceleratedGradientDescent(max_iter=S, gamma=gamma0, initial_step_size=1e-3, max_step_size=1e-1, iteration_callback=False, **decay_kw)
        run2 = solver2.start_device_run(f, torch.zeros(m, dtype=tdt, device=device), rank=rank)
        tA, *_ = timed_window(run2, local, comm, w0, fence, elapsed_max, 64)
        tB, lB, kB, xnB, xmsB = timed_window(run2, local, comm, w1 - w0, fence, elapsed_max, stride)
        tC, *_ = timed_window(run2, local, comm, S - w1, fence, elapsed_max, 64)
        res2 = run2.finish()
        run2.close()
        lam_late = res2.dual_val
        avgB, achB, algB = roof(kB, lB)
        late = {"iterations": [w0 + 1, w1], "ms_per_step": tB / (w1 - w0) * 1e3, "kernel_avg_ms": avgB * 1e3, "achieved_GBps": achB, "frac": achB / HBM_PEAK_GBS,
                "algorithmic_GBps": algB, "algorithmic_frac": algB / HBM_PEAK_GBS}
        if xnB:
            late["exchange_us"] = xmsB / xnB * 1e3
        gamma_end = float(solver2.gamma)
        whole = {"iterations": S, "gamma_continuation": bool(args.gamma_decay), "gamma_first": gamma0, "gamma_last": gamma_end, "seconds": tA + tB + tC, "iterations_per_s": S / (tA + tB + tC), "final_dual_objective": res2.dual_objective,
                 "physical_GBps": roof_bytes * S / (tA + tB + tC) / 1e9, "frac": roof_bytes * S / (tA + tB + tC) / 1e9 / HBM_PEAK_GBS,
                 "algorithmic_GBps": alg_bytes * S / (tA + tB + tC) / 1e9}

    verified = None
    if not args.no_verify:
        try:
            if whole is not None and args.gamma_decay:
                args.gamma = whole["gamma_last"]  # (the verification leg evaluates the objective at the solve's final gamma)
            from benchmark.verify import verify_at_size  # (the checker: oracle slabs, fp64 recomputation, exchange cross-checks)

            verified = verify_at_size(args.dtype, args.gamma, inp, pm_local, f, local, lam_late, rank, world, sharded, device, comm_backend=args.comm)
        except Exception as exc:  # a failed check must show in the line, not kill the measurement
            verified = {"ok": False, "error": f"{type(exc).__name__}: {exc}"}
        if sharded:
            okt = torch.tensor([1.0 if verified.get("ok") else 0.0], dtype=torch.float64, device=device)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            verified["ok_all_ranks"] = bool(okt.item() > 0.5)

    # ---- N > 1: the OTHER split of the entities, timed beside the one the line is quoted on ---------------------------------
    # (the reference cuts contiguously at n // W (+1), dist_utils.py:53-57; the default here gives every rank its share of every
    #  projection block.  A scaling curve quoted on one of them carries the other's per-rank time next to it.)
    compared = None
    if sharded and (world > 1 or emu) and not args.no_partition_compare:
        other = "reference" if args.partition != "reference" else "balanced"
        # Rank-LOCAL work first (cutting this rank's shard can fail on one rank only: memory, an empty block), then the ranks AGREE that all of
        # them got through it before anyone enters the collective part -- a rank that skipped the window alone would leave the others inside
        # the communicator's first collective for ever.  A failure must show in the line, not cost the headline number.
        local_err = None
        try:
            bi2, nnz2, b2, ranges2 = make_shard(other)
            for bi in bi2:
                bi.b_vec = None
        except Exception as exc:
            local_err = f"{type(exc).__name__}: {exc}"
        if world > 1:
            okt = torch.tensor([0.0 if local_err else 1.0], dtype=torch.float64, device=device)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            if float(okt.item()) < 1.0 and local_err is None:
                local_err = "another rank could not cut its shard of this split"
        try:
            if local_err:
                raise RuntimeError(local_err)
            f2 = MatchingSolverDualObjectiveFunctionDistributed(bi2 if nb > 1 else bi2[0], b2, args.gamma, host_device=device, comm_backend=args.comm)
            comm2 = f2.communicator()
            if emu and comm2 is not None:
                comm2.set_emulation(float(emu))
            el2, ln2, kms2, xn2, xms2, _res2 = headline(f2, f2.local_objective, comm2)
            compared = {"kind": other, "ms_per_step": el2 / args.steps * 1e3, "iterations_per_s": args.steps / el2, "kernel_avg_ms": kms2 / max(ln2, 1),
                        "this_rank_columns": [list(r) for r in ranges2], "this_rank_nnz": int(nnz2), "us_per_exchange": (xms2 / xn2 * 1e3) if xn2 else None,
                        "backend": comm2.backend if comm2 is not None else "torch.distributed", "window": [args.warmup + 1, args.warmup + args.steps]}
            if comm2 is not None:
                comm2.close()
            del f2, bi2, b2
        except Exception as exc:
            compared = {"kind": other, "error": f"{type(exc).__name__}: {exc}"}

    # ---- footprint: the handle self-contained, the caller's CSC tensors gone -------------------------------------------------
    footprint = None
    if args.release_inputs and not sharded:
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated(device)
        rel = f.release_inputs()
        inp.A = inp.c = None
        del block_inputs[:]
        gc.collect()
        torch.cuda.synchronize()
        after = torch.cuda.memory_allocated(device)
        el3, ln3, kms3, _, _, _res3 = headline(f, local, comm)
        footprint = {"torch_allocated_before": before, "torch_allocated_after": after, "handle_owned_bytes": rel["owned_bytes"], "kept_elements_of_the_csc_arrays": rel["kept_elements"],
                     "streamed_bytes_per_launch": phys_bytes, "owned_over_streamed": rel["owned_bytes"] / phys_bytes,
                     "note": "torch_allocated counts the generator's CSC tensors (values + int64 indices) and solver state, not the handle's hipMalloc'ed memory (handle_owned_bytes)",
                     "ms_per_step_after_release": el3 / args.steps * 1e3, "kernel_avg_ms_after_release": kms3 / max(ln3, 1), "ms_per_step_before": elapsed / args.steps * 1e3}

    # A launch whose bytes fit the 256 MB Infinity Cache is not bound by HBM at all (BASELINE config 2: 1M entities = 120 MB): its
    # yardstick is what a plain streaming-read launch reaches over a buffer of THAT size on this box (launch included, best of 20)
    cache_resident = None
    if phys_bytes < (256 << 20) and rank == 0:
        ws = max(1 << 20, int(phys_bytes) // 4096 * 4096)
        ceil_ws = read_ceiling_gbps(device, nbytes=ws, reps=20)
        cache_resident = {"working_set_bytes": ws, "note": "the launch's bytes fit the 256 MB Infinity Cache: the fraction of the HBM peak is not the yardstick; "
                          "read_probe_GBps = dl_measure_read_bandwidth (one launch of the streaming-read probe) over a buffer of the same size, read repeatedly",
                          "read_probe_GBps": ceil_ws, "kernel_GBps": achieved, "kernel_frac_of_probe": achieved / ceil_ws if ceil_ws > 0 else None,
                          "ceiling_launch_us": ws / ceil_ws / 1e3 if ceil_ws > 0 else None, "kernel_launch_us": avg_kernel_s * 1e6}

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
                "parallelism": f"column-shard x{world}" + (f", {nb} blocks per rank" if nb > 1 else ""),
                "partition": args.partition if sharded else None,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "bytes": "traffic (HBM bytes per launch: counters when recorded for this configuration, else the layout's) / kernel_avg_ms",
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": "matching_fused_kernel4" if lay["layout"] == 4 else "matching_fused_kernel",
                "kernel_avg_ms": avg_kernel_s * 1e3,
                "kernel_launches": launches,
                "event_stride": stride,
                "physical_bytes_per_launch": phys_bytes,
                "window": [args.warmup + 1, args.warmup + args.steps],
                "frac_of_read_probe": None,
            },
            "aux": {
                "generate_s": t_gen,
                "setup_s": t_setup,
                "final_dual_objective": result.dual_objective,
                "layout": lay,
                "algorithmic_roofline": {
                    "note": "SURVEY.md 8d's figure: 12 E + 4 n + 16 m bytes per launch (4-byte row indices and column pointers, which this kernel does not read: it "
                            "streams 2-byte indices and no pointers) / kernel_avg_ms; NOT a physical fraction -- it exceeds roofline.frac by algorithmic / physical bytes",
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "achieved_GBps": achieved_alg,
                    "frac_of_peak": achieved_alg / HBM_PEAK_GBS,
                    "whole_iteration_GBps": alg_bytes * args.steps / elapsed / 1e9,
                },
                "partition": {"kind": args.partition if sharded else None, "ranks": vworld,
                              "ms_per_step": {args.partition: elapsed / args.steps * 1e3, **({compared["kind"]: compared.get("ms_per_step")} if compared else {})} if sharded else None,
                              "compared": compared,
                              "cost_model": "columns x dist_utils.PROJECTION_COST (simplex 1.14, point-wise 1.0)",
                              "estimated_imbalance_max_over_mean": {k: v["imbalance"] for k, v in ptable.items()},
                              "cuts": ptable[args.partition]["cuts"] if sharded else None,
                              "this_rank_columns": [list(r) for r in ranges_first]},
                "late": late,
                "whole_solve": whole,
                "whole_solve_its_per_s": whole["iterations_per_s"] if whole else None,
                "verified": verified,
                "collective": collective,
                "copy_ceiling_GBps": copy_ceiling_gbps(device),
                "read_probe_GBps": read_ceiling_gbps(device),
                "traffic_counters": traffic_details,
                "cache_resident": cache_resident,
                "footprint": footprint,
            },
        }
        rc_gbps = out["aux"]["read_probe_GBps"]
        out["roofline"]["frac_of_read_probe"] = achieved / rc_gbps if rc_gbps else None  # (same box, same run: what a plain streaming read reaches)
        fastest = max([achieved] + ([late["achieved_GBps"]] if (late and late.get("achieved_GBps")) else []))
        out["roofline"]["read_probe_beaten_by_kernel"] = bool(rc_gbps and fastest > rc_gbps)  # (true: the probe is a lower bound of the box's read rate, nothing more)
        if sharded:
            # the line proves by itself what took part: which GPUs (UUID / PCI address per rank), which exchange (p2p / p2p-fenced / rccl, whether
            # it degraded mid-run), how far the ranks' kernels were apart (skew) and what one exchange bracket cost on the fastest / slowest rank
            collective = {**(collective or {}), **(identities or {}), "state": (comm.backend if comm is not None else "torch.distributed"),
                          "degrade_happened": bool(comm is not None and comm.degraded), "per_rank": per_rank}
        if comm is not None:
            out["aux"]["collective"] = {**(collective or {}), **comm.info(), "emulated_world": emu or None, "emulated_rank": emu_rank if emu else None,
                                        "exchanges": comm.exchanges, "us_per_exchange": (xms / xn * 1e3) if xn else None,
                                        "bracket": "end of the fused pass -> end of the step's first kernel (slab reduction + exchange + gradient statistics)"}
        elif sharded:
            out["aux"]["collective"] = {**(collective or {}), "backend": "torch.distributed", "fallback_reason": getattr(f, "comm_fallback", None)}
        if world == 1 and not args.no_cpu_baseline and inp.A is not None:
            inp.b_vec = b_vec
            out["cpu_baseline"] = cpu_baseline(args, inp, pm_local, total_nnz)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if sharded:
        dist.barrier(**({} if one_device else {"device_ids": [local_rank]}))
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
