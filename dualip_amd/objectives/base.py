"""Objective-function operator interface (reference: src/dualip/objectives/base.py:8-26)."""
from abc import ABC, abstractmethod
from dataclasses import dataclass

from dualip_amd.types import ObjectiveResult  # noqa: F401  (re-exported like the reference module does)


@dataclass
class BaseInputArgs(ABC):
    """Base of the per-objective input bundles."""

    def __post_init__(self):
        pass


class BaseObjective(ABC):
    """An objective exposes ``calculate(dual_val, gamma=None, save_primal=False, **kwargs) -> ObjectiveResult`` and the
    attribute ``equality_mask`` (read by the maximizer, reference optimizers/agd.py:147)."""

    @abstractmethod
    def calculate(self) -> ObjectiveResult:
        ...
