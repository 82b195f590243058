#!/usr/bin/env python3
"""Scaling sweep of the matching benchmark: GPU counts x problem sizes, one table.

Counterpart of the reference's benchmark/run_scaling_benchmark.py:33-55 (its grid: 25M..250M sources x {1,2,3,4} GPUs x 10 000
iterations, one subprocess per cell, CSV out).  Here every cell is one ``bench.py`` run -- ``python bench.py`` for one GPU,
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` for N -- and the table carries, per cell: iterations/s
(the timed window and the whole 1000-iteration solve), ms per iteration, the fused kernel's fraction of the HBM roofline per
rank (early and late window), the exchange back-end and its cost, the verification verdict, and the speed-up over the one-GPU
cell of the same size.

    python benchmark/run_scaling_benchmark.py                       # N = 1, 2, 4, 8 (those the node has) x 100M entities
    python benchmark/run_scaling_benchmark.py --sizes 25000000 100000000 --gpus 1 2 4 8 --out scaling
    python benchmark/run_scaling_benchmark.py --emulate             # ONE GPU: per-rank cost of an N-GPU run (rank 0's shard, the
                                                                    # exchanged sums scaled by N); rows are labelled "emulated"

Writes <out>.json (every bench line) and <out>.md (the table).
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_cell(n_gpus, entities, args, emulate):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--entities", str(entities), "--proj", args.proj, "--steps", str(args.steps), "--warmup", str(args.warmup)] + args.bench_args
    if n_gpus > 1 or emulate:
        common.append("--no-cpu-baseline")
    if emulate and n_gpus > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(args.port),
               os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-sharded", "--emulate-world", str(n_gpus), "--no-verify"] + common
    elif n_gpus > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1", "--master-port", str(args.port),
               os.path.join(ROOT, "bench.py"), "--gpus", str(n_gpus)] + common
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common
    print("+", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if r.returncode != 0 or line is None:
        return {"error": f"rc={r.returncode}", "stderr_tail": r.stderr[-2000:]}
    return json.loads(line)


def row_of(n_gpus, entities, d, emulate):
    if "error" in d:
        return {"gpus": n_gpus, "entities": entities, "error": d["error"]}
    a, r = d["aux"], d["roofline"]
    late, whole, coll, ver = a.get("late") or {}, a.get("whole_solve") or {}, a.get("collective") or {}, a.get("verified")
    return {
        "gpus": n_gpus, "entities": entities, "emulated": bool(emulate and n_gpus > 1),
        "its_per_s": d["value"], "ms_per_step": d["ms_per_step"], "kernel_ms": r["kernel_avg_ms"], "frac_per_rank": r["frac"],
        "late_ms_per_step": late.get("ms_per_step"), "late_frac_per_rank": late.get("frac"),
        "whole_solve_its_per_s": whole.get("iterations_per_s"), "final_dual_objective": whole.get("final_dual_objective", a.get("final_dual_objective")),
        "collective": coll.get("backend"), "exchange_us": coll.get("us_per_exchange"), "verified": None if ver is None else bool(ver.get("ok_all_ranks", ver.get("ok"))),
        "partition": (a.get("partition") or {}).get("kind"), "ms_per_step_by_partition": (a.get("partition") or {}).get("ms_per_step"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--sizes", type=int, nargs="+", default=[100_000_000])
    ap.add_argument("--proj", default="mixed")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reference-sweep", action="store_true", help="the reference's own grid (benchmark/run_scaling_benchmark.py:33-55): 25M .. 250M sources in steps of 25M x "
                    "{1, 2, 3, 4} GPUs, on the reference's contiguous n // W (+1) split (bench.py --partition reference; the line carries the balanced split's "
                    "per-rank time beside it).  250M entities on ONE GPU need ~60 GB for the generator's tensors: fits the 288 GB of an MI355X")
    ap.add_argument("--emulate", action="store_true", help="one GPU: emulate the per-rank cost of every N > 1 (rows labelled emulated; NOT results)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "scaling"))
    ap.add_argument("--port", type=int, default=29561)
    ap.add_argument("bench_args", nargs="*", help="further bench.py arguments after --")
    args = ap.parse_args()
    import torch

    if args.reference_sweep:
        args.sizes = list(range(25_000_000, 250_000_001, 25_000_000))
        args.gpus = [1, 2, 3, 4]
        args.bench_args = list(args.bench_args) + ["--partition", "reference"]
    have = torch.cuda.device_count()
    rows, lines = [], []
    for entities in args.sizes:
        for n in args.gpus:
            if n > have and not args.emulate:
                rows.append({"gpus": n, "entities": entities, "error": f"node has {have} GPU(s)"})
                continue
            d = run_cell(n, entities, args, args.emulate)
            lines.append({"gpus": n, "entities": entities, "emulated": bool(args.emulate and n > 1), "line": d})
            rows.append(row_of(n, entities, d, args.emulate))
    base = {r["entities"]: r for r in rows if r.get("gpus") == 1 and "error" not in r}
    for r in rows:
        b = base.get(r["entities"])
        if b and "error" not in r:
            r["speedup"] = r["its_per_s"] / b["its_per_s"]
            if r.get("whole_solve_its_per_s") and b.get("whole_solve_its_per_s"):
                r["whole_solve_speedup"] = r["whole_solve_its_per_s"] / b["whole_solve_its_per_s"]
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump({"rows": rows, "lines": lines}, open(args.out + ".json", "w"), indent=1)
    f = lambda v, fmt: "-" if v is None else format(v, fmt)  # noqa: E731
    md = ["| entities | GPUs | it/s (window) | ms/it | speed-up | kernel frac of 8 TB/s per rank (early / late) | whole solve it/s (speed-up) | exchange | verified | ms/it by split |", "|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        if "error" in r:
            md.append(f"| {r['entities']:,} | {r['gpus']} | {r['error']} | | | | | | | |")
            continue
        tag = f"{r['gpus']} (emulated: per-rank cost only)" if r["emulated"] else str(r["gpus"])
        md.append(f"| {r['entities']:,} | {tag} | {r['its_per_s']:.0f} | {r['ms_per_step']:.4f} | {f(r.get('speedup'), '.2f')} | {r['frac_per_rank']:.3f} / {f(r.get('late_frac_per_rank'), '.3f')} | "
                  f"{f(r.get('whole_solve_its_per_s'), '.0f')} ({f(r.get('whole_solve_speedup'), '.2f')}) | {r.get('collective') or '-'} {f(r.get('exchange_us'), '.1f')} us | {r.get('verified')} | "
                  f"{', '.join(f'{k} {v:.4f}' for k, v in (r.get('ms_per_step_by_partition') or {}).items() if v is not None) or '-'} |")
    open(args.out + ".md", "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
