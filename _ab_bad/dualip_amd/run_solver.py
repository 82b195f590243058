"""Solver entry point (reference: src/dualip/run_solver.py:17-146).

``run_solver(input_args, solver_args, compute_args, objective_args)`` moves the inputs to the host device, builds
the objective, runs the accelerated gradient ascent and returns a SolverResult.  ``compute_device_num > 1`` is the
one-process-per-GPU mode: call it from every rank of an initialised ``torch.distributed`` group with the GLOBAL
problem; each rank keeps its own column shard (the reference's multi-device branch cannot be constructed,
run_solver.py:60-67, so this is the intended behaviour rather than a copy of it).
"""
import dataclasses
import os
from dataclasses import fields
from typing import Optional

import torch
import torch.distributed as dist

from dualip_amd.objectives.base import BaseInputArgs
from dualip_amd.objectives.miplib import MIPLIB2017ObjectiveFunction
from dualip_amd.objectives.matching import (
    MatchingInputArgs,
    MatchingSolverDualObjectiveFunction,
    MatchingSolverDualObjectiveFunctionDistributed,
)
from dualip_amd.optimizers.agd import AcceleratedGradientDescent
from dualip_amd.types import ComputeArgs, ObjectiveArgs, SolverArgs, SolverResult
from dualip_amd.utils.mlflow_utils import MLflowConfig, log_hyperparameters, mlflow_run_context
from dualip_amd.utils.dist_utils import contiguous_cuts, global_to_local_projection_map, projection_cost_blocks


def transfer_tensors_to_device(input_args: BaseInputArgs, device: str):
    """New instance of the same dataclass with every tensor field moved to ``device`` (run_solver.py:17-32)."""
    return input_args.to(device)


def _local_shard(input_args: MatchingInputArgs, rank: int, world: int, device, partition: str = "reference") -> MatchingInputArgs:
    """This rank's contiguous column block of the global problem.  ``partition="reference"``: sizes as
    dist_utils.split_tensors_to_devices, n // W (+1 for the first n % W ranks); ``"cost"``: contiguous cuts that equalise the
    ranks' estimated cost when the projection map is made of contiguous blocks of different operators
    (dist_utils.contiguous_cuts).  Only this block is sliced and moved -- cutting all W blocks on every rank would hold the
    whole problem twice per rank before the solve starts."""
    A, c = input_args.A, input_args.c
    n = int(A.size(1))
    if partition not in ("reference", "cost"):
        raise ValueError(f"partition must be 'reference' or 'cost', got {partition}")
    cuts = contiguous_cuts(n, world, projection_cost_blocks(input_args.projection_map) if partition == "cost" else ())
    lo, hi = cuts[rank], cuts[rank + 1]
    colptr = A.ccol_indices()
    k0, k1 = (int(v) for v in colptr[torch.tensor([lo, hi], device=colptr.device)].tolist())
    sub_ptr = (colptr[lo : hi + 1] - k0).to(device)
    rows = A.row_indices()[k0:k1].to(device)
    return MatchingInputArgs(
        A=torch.sparse_csc_tensor(sub_ptr, rows, A.values()[k0:k1].to(device), size=(A.size(0), hi - lo)),
        c=torch.sparse_csc_tensor(sub_ptr, rows, c.values()[k0:k1].to(device), size=(A.size(0), hi - lo)),
        projection_map=global_to_local_projection_map(input_args.projection_map, range(lo, hi)),
        b_vec=None,
        equality_mask=input_args.equality_mask,
    )


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
        local = _local_shard(input_args, rank, world, device, os.environ.get("DUALIP_PARTITION", getattr(compute_args, "partition", "reference")))
        return MatchingSolverDualObjectiveFunctionDistributed(
            local_matching_input_args=local, b_vec=input_args.b_vec, gamma=solver_args.gamma, host_device=compute_args.host_device, use_jacobi_precondition=jac
        )
    if kind == "miplib2017":
        kwargs = dict(objective_args.objective_kwargs or {})
        if objective_args.use_jacobi_precondition:
            kwargs.setdefault("use_jacobi_precondition", True)
        return MIPLIB2017ObjectiveFunction(miplib_input_args=input_args, **kwargs)
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
    if not sharded:
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
    return result
