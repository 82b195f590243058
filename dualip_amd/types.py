"""Argument / result records of the solver API.

Field names, order and defaults are the reference's (src/dualip/types.py:7-50): callers construct these records
positionally and by keyword, so they are part of the drop-in surface.  What is added here belongs to this
implementation: the layout of the device-side iteration log that an ObjectiveResult can be rebuilt from, and small
accessors used by the maximizer and the benchmark drivers.
"""
from dataclasses import dataclass
from typing import Any, Dict, Literal, Optional, Sequence

import torch

# columns of one row of the device-resident iteration log (dl_agd_read_log, include/dualip_hip.h)
LOG_DUAL_OBJECTIVE, LOG_STEP_SIZE, LOG_REG_PENALTY, LOG_DUAL_TIMES_GRAD, LOG_MAX_POS_SLACK, LOG_SUM_POS_SLACK, LOG_GRAD_NORM, LOG_PRIMAL_OBJECTIVE = range(8)


@dataclass
class SolverArgs:
    """Maximizer settings (consumed by run_solver -> AcceleratedGradientDescent)."""

    max_iter: int = 10000                              # iterations of the maximiser (no early stop)
    initial_step_size: float = 1e-5                    # used until 14 Lipschitz estimates exist
    gamma: float = 1e-3                                # ridge weight of the smoothed LP
    max_step_size: float = 0.1                         # cap of 1 / L_max
    initial_dual_path: Optional[str] = None            # torch.save'd dual vector to warm-start from
    gamma_decay_type: Optional[Literal["step"]] = None
    gamma_decay_params: Optional[dict] = None          # {"decay_steps": int, "decay_factor": float}
    save_primal: bool = False                          # keep x of the LAST iteration

    def decay_schedule(self):
        """(decay_steps, decay_factor) of the gamma continuation, or (0, 1.0) when there is none."""
        if self.gamma_decay_type is None:
            return 0, 1.0
        p = self.gamma_decay_params or {}
        return int(p["decay_steps"]), float(p["decay_factor"])

    def final_gamma(self) -> float:
        """gamma after ``max_iter`` iterations of the continuation (every ``decay_steps``-th iteration multiplies it)."""
        steps, factor = self.decay_schedule()
        return self.gamma * factor ** (self.max_iter // steps) if steps > 0 else self.gamma


@dataclass
class ComputeArgs:
    """Where to run: ``host_device`` is a torch device string of a ROCm GPU; ``compute_device_num`` > 1 selects the
    one-process-per-GPU column-sharded objective (call run_solver from every rank of an initialised process group)."""

    host_device: str                # e.g. "cuda:0"
    compute_device_num: int = 1     # ranks of the process group (one per GPU)
    partition: str = "reference"    # (not in the reference) how run_solver cuts the entities over the ranks: "reference" = n // W (+1)
                                    # contiguous blocks; "cost" = contiguous blocks of equal estimated cost; "balanced" = every rank
                                    # takes its share of every block of the projection map (DUALIP_PARTITION overrides)

    @property
    def sharded(self) -> bool:
        return self.compute_device_num > 1


@dataclass
class ObjectiveArgs:
    """Which objective run_solver builds and with what extra constructor arguments."""

    objective_type: Literal["miplib2017", "matching"]   # anything else: ValueError in build_objective
    use_jacobi_precondition: bool = False               # row-normalise A and b before the solve
    objective_kwargs: Optional[Dict[str, Any]] = None   # passed to the objective's constructor


@dataclass
class ObjectiveResult:
    """One evaluation of the dual objective.  ``primal_var`` aliases a buffer owned by the objective (it is overwritten
    by the next ``calculate(save_primal=True)``), as in the reference."""

    dual_gradient: torch.Tensor                          # A x - b, [m]
    dual_objective: torch.Tensor                         # c.x + reg + lambda.(A x - b), 0-dim
    reg_penalty: Optional[torch.Tensor] = None           # gamma / 2 * ||x||^2
    primal_objective: Optional[torch.Tensor] = None      # c.x (with save_primal)
    primal_var: Optional[torch.Tensor] = None            # x, one entry per stored non-zero (with save_primal)
    dual_val_times_grad: Optional[torch.Tensor] = None   # lambda.(A x - b)
    max_pos_slack: Optional[torch.Tensor] = None         # max(max_i grad_i, 0)
    sum_pos_slack: Optional[torch.Tensor] = None         # sum_i max(grad_i, 0)

    @classmethod
    def from_log_row(cls, row: Sequence[float], dual_gradient: torch.Tensor, dtype: torch.dtype, device) -> "ObjectiveResult":
        """Rebuild the scalars of an iteration from one row of the device log (0-dim tensors in the working precision)."""
        t = torch.tensor(list(row), dtype=torch.float64, device=device).to(dtype)
        return cls(
            dual_gradient=dual_gradient,
            dual_objective=t[LOG_DUAL_OBJECTIVE],
            reg_penalty=t[LOG_REG_PENALTY],
            dual_val_times_grad=t[LOG_DUAL_TIMES_GRAD],
            max_pos_slack=t[LOG_MAX_POS_SLACK],
            sum_pos_slack=t[LOG_SUM_POS_SLACK],
        )


@dataclass
class SolverResult:
    """What ``maximize`` returns: the last dual iterate y, the last logged dual objective, the last ObjectiveResult and
    the per-iteration logs (Python floats)."""

    dual_val: torch.Tensor              # y of the last iteration
    dual_objective: float               # last entry of dual_objective_log
    objective_result: ObjectiveResult   # evaluation at the last x
    dual_objective_log: list            # one float per iteration
    step_size_log: list                 # one float per iteration

    @property
    def iterations(self) -> int:
        return len(self.dual_objective_log)
