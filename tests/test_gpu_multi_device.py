"""FIRST CONTACT with more than one physical GPU (``-m gpu``; every test skips on a one-GPU box).

No build session ever had a multi-GPU node: the exchange of the column-sharded iteration has run between
PROCESSES sharing one device only.  What that cannot show -- hipIpc mappings between DIFFERENT devices, peer access, the unfenced store
ordering of the P2P mailboxes over xGMI, RCCL with N > 1 devices -- is exactly what these tests execute, one process per GPU, rank r on
``cuda:r``, a ``nccl`` (= RCCL) process group as the side channel like the reference's distributed driver
(benchmark/run_matching_benchmark_dist.py:35-41).  ``tools/first_contact.sh`` runs them, in this order, before anything is timed on
such a node:

  1. ``dl_allreduce_sum`` on every back-end (p2p, p2p-fenced, rccl): exact integer sums under uneven load between the ranks;
  2. the sharded C loop (``dl_agd_run_matching_sharded``) against the reference's own 2-rank gloo golden trace, ranks bit-identical.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.helpers import gather_results

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need(world):
    if torch.cuda.device_count() < world:
        pytest.skip(f"{torch.cuda.device_count()} GPU(s) on this box, {world} needed: runs on the first multi-GPU lease (tools/first_contact.sh)")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist

    torch.cuda.set_device(rank)  # one process per GPU
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    return dist, dev


def _allreduce_worker(rank, world, port, backend, q):
    dist, dev = _init(rank, world, port)
    try:
        from dualip_amd.utils.comm import Communicator

        n = 10_002
        comm = Communicator(n, dev, backend=backend)
        info = comm.info()
        assert comm.backend == backend and info["world"] == world and info["distinct_devices"] == world, info
        if backend == "rccl":
            assert info["rccl_reported_world"] == world and info["rccl_reported_rank"] == rank, info
        g = torch.Generator(device=dev).manual_seed(1234)  # the same stream of values on every rank
        bad = 0
        burn = torch.empty(1 << 24, device=dev)
        for rnd in range(200):
            parts = torch.randint(-1000, 1000, (world, n), generator=g, device=dev).double()  # integers: every order of summation is exact
            want = parts.sum(0)
            if (rnd + rank) % 3 == 0:  # uneven load: this rank reaches the exchange late
                for _ in range(1 + (rnd % 4)):
                    burn.normal_()
            v = parts[rank].clone()
            comm.all_reduce_(v)
            bad += int((v != want).sum())
        comm.check()
        q.put((rank, bad, comm.exchanges, str(torch.cuda.get_device_properties(dev).uuid)))
        dist.barrier(device_ids=[rank])
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("backend", ["p2p", "p2p-fenced", "rccl"])
def test_allreduce_across_two_devices_is_exact_under_uneven_load(backend, world):
    """``..._two_devices`` in the name: what tools/first_contact.sh selects with ``-k two_devices`` (worlds 4 and 8 run when the node has them)."""
    _need(world)
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_allreduce_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = gather_results(procs, q)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert len({uuid for *_, uuid in got}) == world  # the ranks really sat on `world` different GPUs
    for rank, bad, exchanges, _ in got:
        assert bad == 0, f"rank {rank}: {bad} wrong words"
        assert exchanges >= 200


def _loop_worker(rank, world, port, kind, backend, q):
    dist, dev = _init(rank, world, port)
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
        if kind == "mixed":
            half = int(z["mixed_boundary"])
            pm = {**create_projection_map("box", {"lower": 0.0, "upper": 1.0}, n, indices=range(0, half)), **create_projection_map("simplex", {"z": 1.0}, n, indices=range(half, n))}
            ranges = balanced_block_ranges([(0, half), (half, n)], world, rank)
        else:
            pm = create_projection_map("simplex", {"z": 1.0}, n)
            ranges = balanced_block_ranges([(0, n)], world, rank)
        parts = [sub_problem(p, lo, hi) for lo, hi in ranges]
        colptr = [np.zeros(1, dtype=np.int64)]
        for q_ in parts:
            colptr.append(q_["colptr"][1:] + colptr[-1][-1])
        local = dict(m=p["m"], n=sum(q_["n"] for q_ in parts), colptr=np.concatenate(colptr), rowidx=np.concatenate([q_["rowidx"] for q_ in parts]),
                     a=np.concatenate([q_["a"] for q_ in parts]), c=np.concatenate([q_["c"] for q_ in parts]), b=p["b"])
        local_pm = global_to_local_projection_map(pm, [c for lo, hi in ranges for c in range(lo, hi)])
        args = torch_args(local, "f64", local_pm, str(dev), with_b=False)
        f = MatchingSolverDualObjectiveFunctionDistributed(args, torch.from_numpy(p["b"]), float(gamma), host_device=str(dev), comm_backend=backend)
        solver = AcceleratedGradientDescent(max_iter=int(iters), gamma=float(gamma), initial_step_size=float(s0), max_step_size=float(s1), iteration_callback=False)
        run = solver.start_device_run(f, torch.zeros(p["m"], dtype=torch.float64, device=dev), rank=rank)
        assert run.native_sharded  # the loop, exchange included, runs inside the C library
        run.advance(int(iters))
        res = run.finish()
        run.close()
        comm = f.communicator()
        q.put((rank, np.array(res.dual_objective_log), res.dual_val.cpu().numpy(), comm.backend, comm.info()["distinct_devices"]))
        dist.barrier(device_ids=[rank])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,world,backend", [("simplex", 2, "p2p"), ("mixed", 2, "p2p"), ("simplex", 2, "rccl"), ("simplex", 4, "p2p"), ("mixed", 8, "p2p"), ("simplex", 8, "rccl")])
def test_sharded_c_loop_golden_across_two_devices(kind, world, backend):
    """The reference's own 2- / 4- / 8-rank distributed traces (fixture G3, produced by its gloo run) through the sharded C loop with one rank per
    PHYSICAL GPU: every rank bit-identical, no broadcast."""
    _need(world)
    from tests.helpers import load, relerr

    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, world, port, kind, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    for rank, log, dual, be, distinct in gather_results(procs, q):
        out[rank] = (log, dual, be, distinct)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    z = load("g3_syn2000.npz")
    key = f"simplex1|w{world}|f64" if kind == "simplex" else f"mixed|w{world}|f64"
    want_log, want_dual = z[f"{key}|dual_obj_log"], z[f"{key}|dual_val"]
    for r in range(world):
        assert out[r][2] == backend and out[r][3] == world
        assert np.array_equal(out[0][1], out[r][1]) and np.array_equal(out[0][0], out[r][0])
    assert relerr(out[0][0][:40], want_log[:40]) < 1e-9 and relerr(out[0][0], want_log) < 1e-6 and relerr(out[0][1], want_dual) < 1e-5
