"""Two ranks on ONE GPU (``-m gpu``): the N > 1 route of the maximiser with the real HIP local pass on every rank.

RCCL does not put two ranks on the same device, so the process group is gloo (it all-reduces device tensors through
host staging); everything else is what an 8-GPU run executes per rank: column shard -> fused pass -> slab reduction ->
ONE sum-all-reduce of [A x | c.x | sum x^2] -> the identical device-side AGD step on both ranks.  Expected values: the
golden traces the reference's own distributed objective produced under gloo (tests/golden/g3_syn2000.npz), for the plain
simplex map and for the mixed map split at the key boundary.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.helpers import gather_results, retry_once_if_stalled

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, kind, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunctionDistributed
        from dualip_amd.optimizers.agd import AcceleratedGradientDescent
        from dualip_amd.projections import create_projection_map
        from dualip_amd.utils.dist_utils import balanced_block_ranges, global_to_local_projection_map
        from tests.helpers import load, problem, sub_problem, torch_args

        z = load("g3_syn2000.npz")
        p = problem(z)
        gamma, iters, s0, s1 = z["params"]
        n = p["n"]
        if kind == "fairness":  # the fairness pair, column-sharded: every rank streams its slice of f; the pair's rows are all-reduced like the others
            from dualip_amd.objectives.matching_fairness import MatchingFairnessDualObjectiveFunction

            zf = load("gf_fairness.npz")
            p1 = problem(load("g1_syn2000.npz"))
            n1, m1 = p1["n"], p1["m"]
            lo, hi = (0, n1 // 2) if rank == 0 else (n1 // 2, n1)
            sub = sub_problem(p1, lo, hi)
            k0, k1 = int(p1["colptr"][lo]), int(p1["colptr"][hi])
            b_full = torch.cat([torch.from_numpy(p1["b"]), torch.tensor([float(zf["delta"])] * 2, dtype=torch.float64)])
            largs = torch_args(sub, "f64", create_projection_map("simplex", {"z": 1.0}, sub["n"]), "cuda:0")
            largs.b_vec = b_full.to("cuda:0")
            ff = MatchingFairnessDualObjectiveFunction(largs, 0.02, A_fairness=torch.from_numpy(zf["f|f64"][k0:k1].copy()).to("cuda:0"), native=True)
            f = MatchingSolverDualObjectiveFunctionDistributed(None, b_full, 0.02, host_device="cuda:0", local_objective=ff.inner)
            solver = AcceleratedGradientDescent(max_iter=60, gamma=0.02, initial_step_size=1e-3, max_step_size=0.1, iteration_callback=False)
            res = solver.maximize(f, torch.zeros(m1 + 2, dtype=torch.float64, device="cuda:0"), rank=rank)
            q.put((rank, np.array(res.dual_objective_log), res.dual_val.cpu().numpy(), 0))
            return
        if kind == "run_solver":  # the entry point with compute_device_num = 2: every rank passes the GLOBAL problem (CPU tensors)
            from dualip_amd.run_solver import run_solver
            from dualip_amd.types import ComputeArgs, ObjectiveArgs, SolverArgs

            args = torch_args(p, "f64", create_projection_map("simplex", {"z": 1.0}, n), "cpu")
            res = run_solver(args, SolverArgs(max_iter=int(iters), gamma=float(gamma), initial_step_size=float(s0), max_step_size=float(s1)),
                             ComputeArgs(host_device="cuda:0", compute_device_num=2), ObjectiveArgs(objective_type="matching"))
            q.put((rank, np.array(res.dual_objective_log), res.dual_val.cpu().numpy(), 0))
            return
        if kind in ("mixed", "mixed_custom", "custom_one_rank"):
            half = int(z["mixed_boundary"])
            first = ("box", {"lower": 0.0, "upper": 1.0})
            if kind in ("mixed_custom", "custom_one_rank"):  # the box block through a user-registered operator (dense-block route beside the fused kernel)
                from dualip_amd.projections.base import ProjectionOperator, register

                @register("user_unit_clamp")
                class UnitClamp(ProjectionOperator):
                    def __init__(self):
                        pass

                    def __call__(self, x):
                        return x.clamp(0.0, 1.0)

                first = ("user_unit_clamp", {})
            pm = {
                **create_projection_map(first[0], first[1], n, indices=range(0, half)),
                **create_projection_map("simplex", {"z": 1.0}, n, indices=range(half, n)),
            }
            # every rank takes its share of BOTH blocks (bench.py --partition balanced): two column ranges per rank
            ranges = balanced_block_ranges([(0, half), (half, n)], world, rank)
            if kind == "custom_one_rank":  # contiguous split at the key boundary: ONLY rank 0 holds the user-defined operator, so only
                ranges = [(0, half)] if rank == 0 else [(half, n)]  # its shard needs the per-iteration route -- the ranks must still agree on one
        else:
            pm = create_projection_map("simplex", {"z": 1.0}, n)
            ranges = balanced_block_ranges([(0, n)], world, rank)
        # local problem = concatenation of the rank's column ranges
        parts = [sub_problem(p, lo, hi) for lo, hi in ranges]
        colptr = [np.zeros(1, dtype=np.int64)]
        for q_ in parts:
            colptr.append(q_["colptr"][1:] + colptr[-1][-1])
        local = dict(m=p["m"], n=sum(q_["n"] for q_ in parts), colptr=np.concatenate(colptr), rowidx=np.concatenate([q_["rowidx"] for q_ in parts]),
                     a=np.concatenate([q_["a"] for q_ in parts]), c=np.concatenate([q_["c"] for q_ in parts]), b=p["b"])
        cols = [c for lo, hi in ranges for c in range(lo, hi)]
        local_pm = global_to_local_projection_map(pm, cols)
        args = torch_args(local, "f64", local_pm, "cuda:0", with_b=False)
        f = MatchingSolverDualObjectiveFunctionDistributed(args, torch.from_numpy(p["b"]), float(gamma), host_device="cuda:0")
        solver = AcceleratedGradientDescent(max_iter=int(iters), gamma=float(gamma), initial_step_size=float(s0), max_step_size=float(s1), iteration_callback=False)
        res = solver.maximize(f, torch.zeros(p["m"], dtype=torch.float64, device="cuda:0"), rank=rank)
        q.put((rank, np.array(res.dual_objective_log), res.dual_val.cpu().numpy(), f.local_objective.info()["layout"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["simplex", "mixed", "mixed_custom", "custom_one_rank", "run_solver", "fairness", "simplex_w4"])
@retry_once_if_stalled
def test_two_ranks_share_one_gpu(kind):
    from tests.helpers import load, relerr

    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    world = 4 if kind == "simplex_w4" else 2  # (four ranks: the reference's 4-rank golden trace)
    procs = [ctx.Process(target=_worker, args=(r, world, port, "simplex" if kind == "simplex_w4" else kind, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    out = {}
    for rank, log, dual, layout in gather_results(procs, q):
        out[rank] = (log, dual, layout)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])
    if kind == "fairness":  # the single-process trace of the reference's operators (gf_fairness.npz)
        zf = load("gf_fairness.npz")
        assert relerr(out[0][0][:40], zf["trace|simplex1|f64|obj_log"][:40]) < 1e-8
        assert out[0][1][-2] > 0 and out[0][1][-1] == 0
        return
    z = load("g3_syn2000.npz")
    key = "simplex1|w4|f64" if kind == "simplex_w4" else ("simplex1|w2|f64" if kind in ("simplex", "run_solver") else "mixed|w2|f64")  # (the custom clamp is the same projection as box [0, 1])
    want_log, want_dual = z[f"{key}|dual_obj_log"], z[f"{key}|dual_val"]
    # all ranks apply the identical update: identical duals, bit for bit, without a broadcast
    for r in range(1, world):
        assert np.array_equal(out[0][1], out[r][1]) and np.array_equal(out[0][0], out[r][0])
    assert relerr(out[0][0], want_log) < 1e-7, kind
    assert relerr(out[0][1], want_dual) < 1e-6, kind


def _lp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dualip_amd.objectives.miplib import MIPLIB2017ObjectiveFunctionDistributed, MIPLIBInputArgs
        from dualip_amd.optimizers.agd import AcceleratedGradientDescent
        from dualip_amd.projections.base import ProjectionEntry
        from tests.helpers import load, lp_small_entries

        z = load("g6_lp_small.npz")
        m, n = int(z["m"]), int(z["n"])
        lo, hi = (0, n // 2) if rank == 0 else (n // 2, n)  # this rank's variables
        pm = {}
        for k, (kind, params, idx) in enumerate(lp_small_entries(z)):
            local = [int(i) - lo for i in idx if lo <= i < hi]
            if local:
                pm[f"e{k}"] = ProjectionEntry(kind, dict(params), indices=local)
        A = torch.from_numpy(z["A"][:, lo:hi].copy()).to(torch.float64).to_sparse_coo().to("cuda:0")
        args = MIPLIBInputArgs(A=A, c=torch.from_numpy(z["c"][lo:hi].copy()).to("cuda:0"), b_vec=torch.from_numpy(z["b"]).to("cuda:0"),
                               projection_map=pm, equality_mask=torch.from_numpy(z["eq"]).to("cuda:0"))
        f = MIPLIB2017ObjectiveFunctionDistributed(args, gamma=1e-2)
        res = AcceleratedGradientDescent(max_iter=120, gamma=1e-2, initial_step_size=1e-3, max_step_size=0.1, iteration_callback=False).maximize(
            f, torch.zeros(m, dtype=torch.float64, device="cuda:0"), rank=rank)
        q.put((rank, np.array(res.dual_objective_log), res.dual_val.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@retry_once_if_stalled
def test_generic_lp_sharded_by_variables():
    """The generic-LP objective split by variables over two ranks (one GPU, gloo): the trace of the reference's
    single-process run (fixture G6, 40 x 60 LP with equality rows) while round-off has not been amplified yet."""
    from tests.helpers import load, relerr

    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_lp_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    out = {}
    for rank, log, dual in gather_results(procs, q):
        out[rank] = (log, dual)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    z = load("g6_lp_small.npz")
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert relerr(out[0][0], z["trace|plain|f64|obj_log"][:120]) < 1e-8


def _lp_warm_worker(rank, world, port, path, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dualip_amd.objectives.miplib import MIPLIBInputArgs
        from dualip_amd.projections.base import ProjectionEntry
        from dualip_amd.run_solver import run_solver
        from dualip_amd.types import ComputeArgs, ObjectiveArgs, SolverArgs
        from tests.helpers import load, lp_small_entries

        z, zw = load("g6_lp_small.npz"), load("g6_lp_warm.npz")
        n_cold, n_warm, gamma, s0 = zw["small|params"]
        pm = {f"e{k}": ProjectionEntry(kind, dict(params), indices=[int(i) for i in idx]) for k, (kind, params, idx) in enumerate(lp_small_entries(z))}
        # every rank hands over the GLOBAL problem (CPU tensors), as for the matching objective
        args = MIPLIBInputArgs(A=torch.from_numpy(z["A"]).to_sparse_coo(), c=torch.from_numpy(z["c"]), b_vec=torch.from_numpy(z["b"]), projection_map=pm,
                               equality_mask=torch.from_numpy(z["eq"]))
        res = run_solver(args, SolverArgs(max_iter=int(n_warm), gamma=float(gamma), initial_step_size=float(s0), max_step_size=0.1, initial_dual_path=path),
                         ComputeArgs(host_device="cuda:0", compute_device_num=world), ObjectiveArgs(objective_type="miplib2017"))
        q.put((rank, np.array(res.dual_objective_log), res.dual_val.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@retry_once_if_stalled
def test_generic_lp_warm_start_two_ranks_through_run_solver(tmp_path):
    """BASELINE config 5 on more than one GPU, as worded: ``run_solver(objective_type="miplib2017", compute_device_num=2,
    initial_dual_path=...)`` -- the LP sharded by variables (run_solver._local_lp_shard), both ranks warm-started from the duals the
    REFERENCE saved -- walks the reference's single-process warm trace (fixture g6_lp_warm.npz) with bit-identical ranks."""
    from tests.helpers import load, relerr

    zw = load("g6_lp_warm.npz")
    path = str(tmp_path / "ref_dual.pt")
    torch.save(torch.from_numpy(zw["small|f64|cold_lam"]), path)
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_lp_warm_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    out = {}
    for rank, log, dual in gather_results(procs, q):
        out[rank] = (log, dual)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    want = zw["small|f64|warm_obj_log"]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert relerr(out[0][0][:40], want[:40]) < 1e-8 and relerr(out[0][0], want) < 5e-3
    assert out[0][0][0] > zw["small|f64|cold_obj_log"][0]


@retry_once_if_stalled
def test_bench_harness_with_two_ranks_on_one_gpu():
    """bench.py under torch.distributed.run with WORLD_SIZE=2 (developer mode: both ranks on cuda:0, gloo collectives): the
    N > 1 harness -- block-balanced shards, the exchange, max-over-ranks timing -- prints exactly ONE JSON line with the
    contract's fields.  (The number itself means nothing in this mode.)"""
    import json
    import subprocess

    env = dict(os.environ, DUALIP_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--entities", "2000000", "--steps", "4", "--warmup", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "strong" and d["higher_is_better"] is True
    assert d["metric"] == "dual_ascent_iterations_per_sec" and d["value"] > 0 and abs(d["value"] * d["ms_per_step"] - 1000.0) < 1e-6 * 1000
    assert d["config"]["entities"] == 2000000 and d["config"]["parallelism"] == "column-shard x2" and d["cpu_baseline"] is None
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1.2
    # the same global problem on one rank: same non-zeros, same objective after the same iterations
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--entities", "2000000", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                         env=dict(os.environ), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert d1["config"]["nnz"] == d["config"]["nnz"]
    assert abs(d1["aux"]["final_dual_objective"] - d["aux"]["final_dual_objective"]) <= 1e-5 * abs(d1["aux"]["final_dual_objective"])


@retry_once_if_stalled
def test_bench_spawns_its_own_ranks():
    """``python bench.py --gpus 2`` with NO launcher around it (WORLD_SIZE unset): the script re-executes itself under
    torch.distributed.run, and rank 0 prints ONE JSON line that says how the entities were partitioned, which exchange the
    ranks landed on and that it agreed with torch.distributed before anything was timed.  (Developer mode: both ranks on
    cuda:0; the number means nothing.)"""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DUALIP_BENCH_ONE_DEVICE="1", DUALIP_COMM_SOAK_ROUNDS="200")
    for partition in ("contiguous", "balanced"):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--entities", "2000000", "--steps", "4", "--warmup", "2", "--no-late", "--partition", partition]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["config"]["partition"] == partition and d["aux"]["partition"]["kind"] == partition
        coll = d["aux"]["collective"]
        assert coll["backend"] in ("p2p", "p2p-fenced", "rccl", "torch.distributed") and coll["selftest"]["ok"] is True, coll
        # the line is self-proving: who took part (one record per rank: UUID / PCI address / ordinal / pid), on how many distinct GPUs (one here,
        # on purpose, and flagged), through which exchange, how far the ranks' kernels were apart, what one exchange bracket cost per rank
        assert [r["rank"] for r in coll["ranks"]] == [0, 1] and all(r["uuid"] and r["pid"] > 0 and r["device_ordinal"] == 0 for r in coll["ranks"]), coll["ranks"]
        assert coll["ranks"][0]["pid"] != coll["ranks"][1]["pid"] and coll["distinct_gpus"] == 1 and coll["one_device_mode"] is True
        assert coll["process_group"] == {"backend": "gloo", "world_size": 2} and coll["state"] == coll["backend"] and coll["degrade_happened"] is False
        pr = coll["per_rank"]
        assert len(pr["kernel_avg_ms"]) == 2 and 0 < pr["kernel_avg_ms_min"] <= pr["kernel_avg_ms_max"] and pr["kernel_skew"] >= 0 and sum(pr["nnz"]) == d["config"]["nnz"]
        assert len(pr["us_per_exchange"]) == 2 and coll["us_per_exchange"] > 0 and coll["world"] == 2
        assert d["aux"]["verified"]["ok_all_ranks"] is True, d["aux"]["verified"]
        r = d["roofline"]
        assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["traffic"] > 0
        if partition == "contiguous":  # mixed map, cost-weighted cuts: rank 0's range ends past the count-balanced middle... of the box half
            cuts = d["aux"]["partition"]["cuts"]
            assert cuts[0] == 0 and cuts[-1] == 2000000 and 1000000 < cuts[1] < 1100000, cuts


@retry_once_if_stalled
def test_bench_harness_with_eight_ranks_on_one_gpu():
    """The target machine's world size (benchmark/run_matching_benchmark_dist.py:33-193: eight ranks) through the whole harness before
    an 8-GPU node ever runs it: ``bench.py --gpus 8`` spawning its own ranks, all on cuda:0 (developer mode; the number means
    nothing).  The exchange must have passed its soak self-test with EIGHT slots per mailbox, agreed with torch.distributed,
    the line must carry both partitions' records, and every rank's duals must be bit-identical (``verified.ok_all_ranks``)."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DUALIP_BENCH_ONE_DEVICE="1", DUALIP_COMM_SOAK_ROUNDS="200", HSA_ENABLE_SDMA="0")  # (nine processes on one device: no copy-engine queues for the ranks, tests/test_gpu_comm.py)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--entities", "800000", "--steps", "4", "--warmup", "2", "--no-late"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "column-shard x8" and d["config"]["entities"] == 800000
    coll = d["aux"]["collective"]
    assert coll["backend"] in ("p2p", "p2p-fenced") and coll["selftest"]["ok"] is True, coll
    part = d["aux"]["partition"]
    assert part["kind"] == "balanced" and part["ranks"] == 8, part
    cmp_ = part["compared"]  # the library default n // W (+1) (dist_utils.py:53-57) measured beside the benchmark's split
    assert cmp_["kind"] == "reference" and "error" not in cmp_ and cmp_["ms_per_step"] > 0, cmp_
    assert d["aux"]["verified"]["ok_all_ranks"] is True, d["aux"]["verified"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] - 1000.0) < 1e-3
