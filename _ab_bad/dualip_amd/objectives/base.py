"""Operator interface of the objectives (reference: src/dualip/objectives/base.py:8-26).

Two base classes, both part of the API: input bundles derive from ``BaseInputArgs`` (run_solver moves every tensor field
of such a dataclass to the host device), objectives from ``BaseObjective``.  The maximizer needs ``calculate`` and the
attribute ``equality_mask``; an objective that additionally sets ``_dualip_native = True`` and offers
``calculate_packed_ptr(lambda_ptr, gamma)`` -> float64[m + 2] = [A x | c.x | sum x^2] is driven by the device-resident
loop instead of the generic torch one (optimizers/agd.py).
"""
from abc import ABC, abstractmethod
from dataclasses import dataclass, fields, replace

import torch

from dualip_amd.types import ObjectiveResult  # noqa: F401  (re-exported like the reference module does)


@dataclass
class BaseInputArgs(ABC):
    """Marker base of the per-objective input dataclasses (MatchingInputArgs, MIPLIBInputArgs)."""

    def __post_init__(self):
        pass

    def to(self, device) -> "BaseInputArgs":
        """A copy of the record whose tensor fields live on ``device`` (what run_solver does before building the
        objective; non-tensor fields such as the projection map are shared, not copied)."""
        moved = {f.name: getattr(self, f.name).to(device) for f in fields(self) if isinstance(getattr(self, f.name), torch.Tensor)}
        return replace(self, **moved)


class BaseObjective(ABC):
    #: set by native objectives: the maximizer keeps its state on the device and calls calculate_packed_ptr
    _dualip_native = False
    #: native objectives that return the packed float64 [A x | c.x | sum x^2] buffer per call (sharded matching, generic LP,
    #: maps with user-defined operators) instead of running wholly inside dl_agd_run_matching
    _dualip_packed = False
    #: the objective reads the duals with torch ops: hand them over as a tensor (calculate_packed), not as a device address
    _needs_dual_tensor = False
    #: rows of the dual that are equality constraints (bool tensor or None); read by the maximizer (reference agd.py:147)
    equality_mask = None

    @abstractmethod
    def calculate(self) -> ObjectiveResult:
        """``calculate(dual_val, gamma=None, save_primal=False, **kwargs)``: gradient, objective value and penalty at ``dual_val``."""
