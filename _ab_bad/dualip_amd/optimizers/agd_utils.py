"""Step-size rule of the maximiser for the generic (user-defined objective) path.

Reference: src/dualip/optimizers/agd_utils.py:4-89.  The function names and signatures of the reference module are
kept for callers and tests; the native matching path does not use this file -- its step size is computed on the
device by ``agd_stats_kernel`` + ``agd_apply_kernel`` (csrc/agd_kernels.hip) with the same rule.

Rule: keep the last ``max_history_length`` (gradient, dual) pairs; L_j = ||g_{j+1}-g_j|| / ||y_{j+1}-y_j|| for
consecutive pairs; fewer than ``max_history_length - 1`` estimates -> initial_step_size; otherwise
min(1 / max_j L_j, max_step_size), falling back to initial_step_size when the maximum is NaN/inf and to
max_step_size when it is exactly 0.  ``max`` is Python's builtin over 0-dim tensors, oldest first, so a NaN is only
noticed in first position -- that quirk is part of the reference's behaviour and is reproduced.
"""
import math

import torch


def norm_of_difference(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    return torch.linalg.vector_norm(x - y)


def update_dual_gradient_history(gradient, dual_val, grad_history: list, dual_history: list, max_history_length: int) -> None:
    """Append detached copies; both lists are trimmed from the front to ``max_history_length``."""
    while len(grad_history) >= max_history_length:
        del grad_history[0]
        del dual_history[0]
    grad_history.append(gradient.detach().clone())
    dual_history.append(dual_val.detach().clone())


def estimate_lipschitz_constant(grad_one, grad_two, dual_one, dual_two) -> torch.Tensor:
    return norm_of_difference(grad_one, grad_two) / norm_of_difference(dual_one, dual_two)


def _builtin_max(values):
    best = values[0]
    for v in values[1:]:
        if v > best:
            best = v
    return best


def step_size_from_lipschitz_constants(lipschitz_constants: list, max_history_length: int, initial_step_size: float, max_step_size: float) -> float:
    if len(lipschitz_constants) < max(1, max_history_length - 1):
        return initial_step_size
    top = float(_builtin_max(lipschitz_constants))
    if math.isnan(top) or math.isinf(top):
        return initial_step_size
    candidate = max_step_size if top == 0.0 else 1.0 / top
    return min(candidate, max_step_size)


def calculate_step_size(
    dual_grad,
    dual_val,
    grad_history: list,
    dual_history: list,
    max_history_length: int = 15,
    initial_step_size: float = 1e-5,
    max_step_size: float = 0.1,
) -> float:
    update_dual_gradient_history(dual_grad, dual_val, grad_history, dual_history, max_history_length)
    pairs = zip(grad_history[:-1], grad_history[1:], dual_history[:-1], dual_history[1:])
    constants = [estimate_lipschitz_constant(g0, g1, y0, y1) for g0, g1, y0, y1 in pairs]
    return step_size_from_lipschitz_constants(constants, max_history_length, initial_step_size, max_step_size)
