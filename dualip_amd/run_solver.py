"""Solver entry point (reference: src/dualip/run_solver.py:17-146).

``run_solver(input_args, solver_args, compute_args, objective_args)`` moves the inputs to the host device, builds
the objective, runs the accelerated gradient ascent and returns a SolverResult.  ``compute_device_num > 1`` is the
one-process-per-GPU mode: call it from every rank of an initialised ``torch.distributed`` group with the GLOBAL
problem; each rank keeps its own column shard (the reference's multi-device branch cannot be constructed,
run_solver.py:60-67, so this is the intended behaviour rather than a copy of it).
"""
import dataclasses
import os
from typing import Optional

import torch
import torch.distributed as dist

from dualip_amd import _hip
from dualip_amd.objectives.base import BaseInputArgs
from dualip_amd.objectives.miplib import MIPLIB2017ObjectiveFunction, MIPLIB2017ObjectiveFunctionDistributed, MIPLIBInputArgs
from dualip_amd.objectives.matching import (
    MatchingInputArgs,
    MatchingSolverDualObjectiveFunction,
    MatchingSolverDualObjectiveFunctionDistributed,
)
from dualip_amd.optimizers.agd import AcceleratedGradientDescent
from dualip_amd.types import ComputeArgs, ObjectiveArgs, SolverArgs, SolverResult
from dualip_amd.utils.mlflow_utils import MLflowConfig, log_hyperparameters, mlflow_run_context
from dualip_amd.utils.dist_utils import balanced_block_ranges, contiguous_cuts, global_to_local_projection_map, projection_cost_blocks


def transfer_tensors_to_device(input_args: BaseInputArgs, device: str):
    """New instance of the same dataclass with every tensor field moved to ``device`` (run_solver.py:17-32)."""
    return input_args.to(device)


def _column_blocks(projection_map, n: int):
    """The map as contiguous blocks [(lo, hi), ...] covering [0, n) in order -- one per entry whose indices are a contiguous
    ``range``, plus the gaps between them -- or None when an entry's columns are scattered."""
    runs = []
    for entry in projection_map.values():
        idx = entry.indices
        if not (isinstance(idx, range) and idx.step == 1):
            return None
        if len(idx):
            runs.append((idx.start, idx.stop))
    runs.sort()
    blocks, pos = [], 0
    for lo, hi in runs:
        if lo < pos:
            return None
        if lo > pos:
            blocks.append((pos, lo))
        blocks.append((lo, hi))
        pos = hi
    if pos < n:
        blocks.append((pos, n))
    return blocks


def _local_shard(input_args: MatchingInputArgs, rank: int, world: int, device, partition: str = "reference", stage_in_objective: bool = True) -> MatchingInputArgs:
    """This rank's columns of the global problem.

    ``partition="reference"``: one contiguous block, sizes as dist_utils.split_tensors_to_devices -- n // W (+1 for the first
    n % W ranks);  ``"cost"``: one contiguous block, cut so that the ranks' estimated COSTS are equal when the projection map is
    made of contiguous blocks of different operators (dist_utils.contiguous_cuts);  ``"balanced"``: the rank's share of EVERY
    block of the map (dist_utils.balanced_block_ranges) -- not contiguous, but every rank then carries the same operator mix,
    which is what the fused pass likes best: point-wise columns stream at the memory's pace and hide the simplex columns'
    arithmetic inside every CU (measured, profiles/r03_partitions_emulated_1gpu.md: 0.223 ms per iteration against 0.236 ms
    for equal-cost contiguous cuts and 0.2455 ms for the reference's cuts, 100M-entity mixed map on 8 ranks).  The sums of a
    dual-ascent iteration do not depend on which rank holds which column.
    Only this rank's columns are sliced and moved -- cutting all W blocks on every rank would hold the whole problem twice per
    rank before the solve starts."""
    A, c = input_args.A, input_args.c
    n = int(A.size(1))
    if partition not in ("reference", "cost", "balanced"):
        raise ValueError(f"partition must be 'reference', 'cost' or 'balanced', got {partition}")
    blocks = _column_blocks(input_args.projection_map, n) if partition == "balanced" else None
    if partition == "balanced" and blocks is None:
        # (entries given as index LISTS -- create_projection_map turns short ranges into lists -- or overlapping / strided ranges: there are
        #  no contiguous blocks to share out.  Said out loud: a run that asked for one split must not silently be timed on another.)
        import warnings

        warnings.warn("dualip_amd.run_solver: partition='balanced' needs a projection map whose entries are disjoint step-1 ranges; this map is not -- "
                      "using the reference's contiguous cut n // W (+1) instead")
    if blocks is not None and len(blocks) > 1:
        pieces = balanced_block_ranges(blocks, world, rank)
    else:
        cuts = contiguous_cuts(n, world, projection_cost_blocks(input_args.projection_map) if partition == "cost" else ())
        pieces = [(cuts[rank], cuts[rank + 1])]
    colptr = A.ccol_indices()
    if stage_in_objective and not A.values().is_cuda and not c.values().is_cuda:
        # a host-resident problem (the reference's drivers: run_matching_benchmark_dist.py:95-110): the shard is cut on the HOST -- views of the
        # caller's arrays for a contiguous block -- and the objective stages what the kernel needs through dl_stage_to_device (pinned, chunked,
        # row indices narrowed on the way) instead of one pageable torch copy per field here
        device = A.values().device
    ptrs, rows, a_vals, c_vals, cols, off = [torch.zeros(1, dtype=colptr.dtype, device=device)], [], [], [], [], 0
    for lo, hi in pieces:
        k0, k1 = (int(v) for v in colptr[torch.tensor([lo, hi], device=colptr.device)].tolist())
        ptrs.append((colptr[lo + 1 : hi + 1] - k0 + off).to(device))
        rows.append(A.row_indices()[k0:k1].to(device))
        a_vals.append(A.values()[k0:k1].to(device))
        c_vals.append(c.values()[k0:k1].to(device))
        cols.append(range(lo, hi))
        off += k1 - k0
    width = sum(len(r) for r in cols)
    if not rows:  # a rank without columns (blocks narrower than the world): an EMPTY shard -- it still takes part in every exchange
        rows, a_vals, c_vals = [A.row_indices()[:0].to(device)], [A.values()[:0].to(device)], [c.values()[:0].to(device)]
    sub_ptr, sub_rows = torch.cat(ptrs), torch.cat(rows)
    if len(cols) == 1:
        local_map = global_to_local_projection_map(input_args.projection_map, cols[0])
    else:  # several pieces: every entry of the map re-based piece by piece (positions in the concatenated shard)
        local_map, pos = {}, 0
        for piece in cols:
            for key, entry in global_to_local_projection_map(input_args.projection_map, piece).items():
                idx = entry.indices
                shifted = range(idx.start + pos, idx.stop + pos) if isinstance(idx, range) else [i + pos for i in idx]
                if key in local_map:  # (an entry spanning two pieces: keep ONE entry, its columns as a list)
                    shifted = list(local_map[key].indices) + list(shifted)
                local_map[key] = type(entry)(proj_type=entry.proj_type, proj_params=entry.proj_params, indices=shifted)
            pos += len(piece)
    return MatchingInputArgs(
        A=torch.sparse_csc_tensor(sub_ptr, sub_rows, torch.cat(a_vals), size=(A.size(0), width)),
        c=torch.sparse_csc_tensor(sub_ptr, sub_rows, torch.cat(c_vals), size=(A.size(0), width)),
        projection_map=local_map,
        b_vec=None,
        equality_mask=input_args.equality_mask,
    )


def _local_lp_shard(input_args: MIPLIBInputArgs, rank: int, world: int, device) -> MIPLIBInputArgs:
    """This rank's VARIABLES (columns of A, entries of c, projection entries re-based) of a generic LP and the full b_vec -- the
    contiguous cut n // W (+1 for the first n % W ranks) of dist_utils.split_tensors_to_devices (dist_utils.py:53-57), applied to
    the LP's columns.  Any layout of A the objective accepts (dense, COO, CSR, CSC); only this rank's columns are moved."""
    A = input_args.A
    n = int(A.shape[1])
    cuts = contiguous_cuts(n, world, ())
    lo, hi = cuts[rank], cuts[rank + 1]
    if A.layout == torch.strided:
        A_loc = A[:, lo:hi].contiguous()
    else:
        coo = (A if A.layout == torch.sparse_coo else A.to_sparse_coo()).coalesce()
        idx, vals = coo.indices(), coo.values()
        keep = (idx[1] >= lo) & (idx[1] < hi)
        A_loc = torch.sparse_coo_tensor(torch.stack([idx[0][keep], idx[1][keep] - lo]), vals[keep], (int(A.shape[0]), hi - lo)).coalesce()
    eq = input_args.equality_mask
    return MIPLIBInputArgs(A=A_loc.to(device), c=input_args.c[lo:hi].contiguous().to(device), b_vec=input_args.b_vec.to(device),
                           projection_map=global_to_local_projection_map(input_args.projection_map, range(lo, hi)),
                           equality_mask=eq.to(device) if eq is not None else None)


def build_objective(input_args: BaseInputArgs, solver_args: SolverArgs, compute_args: ComputeArgs, objective_args: ObjectiveArgs):
    kind = objective_args.objective_type
    if kind == "matching":
        jac = bool(objective_args.use_jacobi_precondition)
        if compute_args.compute_device_num == 1:
            return MatchingSolverDualObjectiveFunction(matching_input_args=input_args, gamma=solver_args.gamma, use_jacobi_precondition=jac)
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("compute_device_num > 1 needs an initialised torch.distributed group (one process per GPU)")
        rank, world = dist.get_rank(), dist.get_world_size()
        if world != compute_args.compute_device_num:
            raise ValueError(f"compute_device_num={compute_args.compute_device_num} but the process group has {world} ranks")
        device = torch.device("cuda", torch.cuda.current_device())
        partition = os.environ.get("DUALIP_PARTITION", getattr(compute_args, "partition", "reference"))
        kinds = [None] * world  # (ComputeArgs and the environment are per process: every rank must cut the SAME way, or columns are lost / doubled)
        dist.all_gather_object(kinds, partition)
        if len(set(kinds)) != 1:
            raise ValueError(f"the ranks disagree on the partition of the entities ({kinds}): set ComputeArgs.partition / DUALIP_PARTITION identically on every rank")
        local = _local_shard(input_args, rank, world, device, partition, stage_in_objective=not jac)  # (Jacobi scales whole device tensors)
        return MatchingSolverDualObjectiveFunctionDistributed(
            local_matching_input_args=local, b_vec=input_args.b_vec, gamma=solver_args.gamma, host_device=compute_args.host_device, use_jacobi_precondition=jac
        )
    if kind == "miplib2017":
        kwargs = dict(objective_args.objective_kwargs or {})
        if objective_args.use_jacobi_precondition:
            kwargs.setdefault("use_jacobi_precondition", True)
        if compute_args.compute_device_num == 1:
            return MIPLIB2017ObjectiveFunction(miplib_input_args=input_args, **kwargs)
        # BASELINE config 5 on several GPUs: the LP sharded by variables, one process per GPU (the reference's MIPLIB objective is
        # single-device; x_j depends on column j only and A x is a sum over columns, so the shards' packed partials add up exactly)
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("compute_device_num > 1 needs an initialised torch.distributed group (one process per GPU)")
        rank, world = dist.get_rank(), dist.get_world_size()
        if world != compute_args.compute_device_num:
            raise ValueError(f"compute_device_num={compute_args.compute_device_num} but the process group has {world} ranks")
        if kwargs.get("use_jacobi_precondition"):
            raise NotImplementedError("Jacobi preconditioning of the generic-LP objective is single-device (row norms of the whole A)")
        device = torch.device("cuda", torch.cuda.current_device())
        return MIPLIB2017ObjectiveFunctionDistributed(_local_lp_shard(input_args, rank, world, device), gamma=solver_args.gamma)
    raise ValueError(f"Objective type {kind} not supported")


def run_solver(
    input_args: BaseInputArgs,
    solver_args: SolverArgs,
    compute_args: ComputeArgs,
    objective_args: ObjectiveArgs,
    mlflow_config: Optional[MLflowConfig] = None,
) -> SolverResult:
    if mlflow_config is None:
        mlflow_config = MLflowConfig(enabled=False)
    with mlflow_run_context(mlflow_config):
        if mlflow_config.enabled and mlflow_config.log_hyperparameters:  # (run_solver.py:100-105)
            log_hyperparameters({"solver": dataclasses.asdict(solver_args), "objective": dataclasses.asdict(objective_args)})
        return _run_solver(input_args, solver_args, compute_args, objective_args)


def _run_solver(input_args, solver_args, compute_args, objective_args) -> SolverResult:
    host_device = compute_args.host_device
    sharded = compute_args.compute_device_num > 1
    # host_device="cpu" (the default of the reference's examples: movies_lens_matching.py:227, solve_miplib_dataset.py:58) names where the
    # CALLER keeps its tensors, not where the arithmetic runs: the inputs are staged to this process's current ROCm device, the solve is
    # libdualip_hip.so's as always, and the SolverResult comes back on the CPU.  Without a GPU this raises (there is no CPU compute path).
    caller_device = None
    if torch.device(host_device).type == "cpu":
        caller_device = torch.device(host_device)
        host_device = _hip.compute_device()
        _hip.stage(input_args.b_vec, "run_solver inputs (host_device='cpu')", host_device)  # (says it once in the log; raises without a GPU)
        compute_args = dataclasses.replace(compute_args, host_device=str(host_device))
    stages_itself = caller_device is not None and objective_args.objective_type == "matching" and not objective_args.use_jacobi_precondition
    if not sharded and not stages_itself:  # (a host-resident matching problem: the objective stages the arrays the kernel needs -- dl_stage_to_device)
        input_args = transfer_tensors_to_device(input_args, host_device)
    objective = build_objective(input_args, solver_args, compute_args, objective_args)
    solver = AcceleratedGradientDescent(
        initial_step_size=solver_args.initial_step_size,
        max_iter=solver_args.max_iter,
        max_step_size=solver_args.max_step_size,
        gamma=solver_args.gamma,
        gamma_decay_type=solver_args.gamma_decay_type,
        gamma_decay_params=solver_args.gamma_decay_params,
        save_primal=solver_args.save_primal,
    )
    if solver_args.initial_dual_path is not None:
        initial_dual = torch.load(solver_args.initial_dual_path)  # warm start
    else:
        initial_dual = torch.zeros_like(input_args.b_vec)
    device = objective.device if hasattr(objective, "device") else host_device
    initial_dual = initial_dual.to(device)
    rank = dist.get_rank() if (sharded and dist.is_initialized()) else 0
    result = solver.maximize(objective, initial_dual, rank=rank)
    if getattr(objective, "use_jacobi_precondition", None):  # report duals / gradient of the original rows (run_solver.py:136-144)
        dual_val, dual_grad = objective.invert_jacobi_precondition(result.dual_val, result.objective_result.dual_gradient)
        result.dual_val = dual_val
        result.objective_result.dual_gradient = dual_grad
    return _hip.result_to(result, caller_device)
