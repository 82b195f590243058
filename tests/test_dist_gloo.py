"""world_size-2 gloo tests (CPU) of the N>1 path: column sharding helpers + the distributed objective's single
sum-all-reduce + the maximiser's rank-0 update/broadcast.  The per-rank fused pass is replaced by the oracle-backed
stand-in (tests/helpers.py:OracleLocalObjective); expected values are golden traces produced by the reference's own
distributed objective under gloo (tests/golden/g3_syn2000.npz) and the reference's 5x5 known-answer trace
(tests/distributed/test_matching_distributed.py:183-194)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunctionDistributed
        from dualip_amd.optimizers.agd import AcceleratedGradientDescent
        from dualip_amd.projections import create_projection_map
        from dualip_amd.utils.dist_utils import global_to_local_projection_map, split_tensors_to_devices
        from tests.helpers import NP_DT, OracleLocalObjective, load, problem, scala_5x5, sub_problem

        out = {}
        if case["kind"] == "scala":
            p = scala_5x5()
            dn, gamma, iters, s0, s1 = "f32", 1e-3, 30, 1e-5, 0.1
            lam0 = 0.1 * np.ones(5, dtype=np.float32)
            pm = create_projection_map("simplex", {"z": 1}, p["n"])
        else:
            z = load("g3_syn2000.npz")
            p = problem(z)
            dn = case["dtype"]
            gamma, iters, s0, s1 = z["params"]
            iters = int(iters)
            lam0 = np.zeros(p["m"], dtype=NP_DT[dn])
            if case["kind"] == "mixed":
                half = int(z["mixed_boundary"])
                pm = {
                    **create_projection_map("box", {"lower": 0.0, "upper": 1.0}, p["n"], indices=range(0, half)),
                    **create_projection_map("simplex", {"z": 1.0}, p["n"], indices=range(half, p["n"])),
                }
            else:
                pm = create_projection_map("simplex", {"z": 1.0}, p["n"])
        npdt = NP_DT[dn]
        td = torch.float32 if dn == "f32" else torch.float64
        # shard exactly as the product does: contiguous balanced column ranges + re-based projection map
        A = torch.sparse_csc_tensor(torch.from_numpy(p["colptr"]), torch.from_numpy(p["rowidx"]), torch.from_numpy(np.asarray(p["a"], dtype=npdt)), size=(p["m"], p["n"]))
        a_blocks, _, index_map = split_tensors_to_devices(A, A, ["cpu"] * world)
        cols = index_map[rank]
        local_pm = global_to_local_projection_map(pm, cols)
        sub = sub_problem(p, cols.start, cols.stop)
        assert a_blocks[rank].shape[1] == sub["n"] and np.array_equal(a_blocks[rank].ccol_indices().numpy(), sub["colptr"])
        projs, col_proj = [], np.full(sub["n"], -1, dtype=np.int32)
        for pid, (_, e) in enumerate(local_pm.items()):
            projs.append((e.proj_type, e.proj_params))
            col_proj[np.asarray(list(e.indices), dtype=np.int64)] = pid
        local = OracleLocalObjective(sub, projs, gamma, npdt, col_proj=col_proj)
        f = MatchingSolverDualObjectiveFunctionDistributed(None, torch.from_numpy(np.asarray(p["b"], dtype=npdt)), gamma, host_device="cpu", local_objective=local)
        if case["kind"] != "scala":
            r = f.calculate(torch.from_numpy(np.asarray(z["lam"], dtype=npdt)), gamma=gamma, rank=rank)
            out["single_grad"] = r.dual_gradient.numpy().copy()
            out["single_scal"] = np.array([float(r.dual_objective), float(r.reg_penalty), 0.0, float(r.dual_val_times_grad), float(r.max_pos_slack), float(r.sum_pos_slack)])
        solver = AcceleratedGradientDescent(max_iter=iters, gamma=gamma, initial_step_size=s0, max_step_size=s1, iteration_callback=False)
        res = solver.maximize(f, torch.from_numpy(lam0).to(td), rank=rank)
        out["dual_val"] = res.dual_val.numpy().copy()
        out["dual_obj_log"] = np.array(res.dual_objective_log)
        out["step_log"] = np.array(res.step_size_log)
        # every rank must end with the same dual (broadcast) -- gather and compare on rank 0
        gathered = [None] * world
        dist.all_gather_object(gathered, out["dual_val"])
        if rank == 0:
            out["all_dual_vals"] = gathered
            q.put(out)
        with pytest.raises(NotImplementedError):
            f.calculate(torch.from_numpy(lam0).to(td), save_primal=True)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(case, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    out = None
    for pr in procs:
        pr.join(timeout=240)
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    out = q.get()
    return out


def test_two_ranks_reproduce_scala_known_answer():
    from tests.helpers import SCALA_GOLDEN

    out = _run({"kind": "scala"})
    for i, want in SCALA_GOLDEN:
        assert abs(out["dual_obj_log"][i - 1] - want) < 1e-5
    assert np.array_equal(out["all_dual_vals"][0], out["all_dual_vals"][1])


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_two_ranks_match_reference_distributed_trace(dtype):
    from tests.helpers import load, relerr

    z = load("g3_syn2000.npz")
    out = _run({"kind": "simplex", "dtype": dtype})
    key = f"simplex1|w2|{dtype}"
    tol = 1e-9 if dtype == "f64" else 2e-4
    assert relerr(out["single_grad"], z[f"{key}|single_grad"]) < tol
    assert relerr(out["single_scal"][[0, 1, 3, 4, 5]], z[f"{key}|single_scal"][[0, 1, 3, 4, 5]]) < tol
    n = 40 if dtype == "f64" else 15
    assert relerr(out["dual_obj_log"][:n], z[f"{key}|dual_obj_log"][:n]) < (1e-8 if dtype == "f64" else 1e-4)
    assert np.allclose(out["step_log"][:n], z[f"{key}|step_log"][:n], rtol=1e-6 if dtype == "f64" else 1e-2)
    if dtype == "f64":
        assert relerr(out["dual_val"], z[f"{key}|dual_val"]) < 1e-7
    assert np.array_equal(out["all_dual_vals"][0], out["all_dual_vals"][1])


def test_mixed_map_split_over_two_ranks():
    from tests.helpers import load, relerr

    z = load("g3_syn2000.npz")
    out = _run({"kind": "mixed", "dtype": "f64"})
    key = "mixed|w2|f64"
    assert relerr(out["single_grad"], z[f"{key}|single_grad"]) < 1e-9
    assert relerr(out["dual_obj_log"], z[f"{key}|dual_obj_log"]) < 1e-8
    assert relerr(out["dual_val"], z[f"{key}|dual_val"]) < 1e-7


def test_eight_ranks_match_reference_distributed_trace():
    """World 8 = the target node (benchmark/run_matching_benchmark_dist.py:33-193): the reference's own 8-rank gloo trace; the
    contiguous n // W (+1) cut, the re-based maps and the one sum-all-reduce with eight contributions."""
    from tests.helpers import load, relerr

    z = load("g3_syn2000.npz")
    out = _run({"kind": "simplex", "dtype": "f64"}, world=8)
    key = "simplex1|w8|f64"
    assert relerr(out["single_grad"], z[f"{key}|single_grad"]) < 1e-9
    assert relerr(out["single_scal"][[0, 1, 3, 4, 5]], z[f"{key}|single_scal"][[0, 1, 3, 4, 5]]) < 1e-9
    assert relerr(out["dual_obj_log"], z[f"{key}|dual_obj_log"]) < 1e-8
    assert np.allclose(out["step_log"], z[f"{key}|step_log"], rtol=1e-6)
    assert relerr(out["dual_val"], z[f"{key}|dual_val"]) < 1e-7
    assert len(out["all_dual_vals"]) == 8 and all(np.array_equal(out["all_dual_vals"][0], v) for v in out["all_dual_vals"][1:])
