"""Simplex projections (reference: src/dualip/projections/simplex.py:239-274).

``simplex``     {x >= 0, sum x <= z}  (registry name of the reference's SimplexIneq)
``simplex_eq``  {x >= 0, sum x == z}

The reference offers two numerical methods.  ``duchi`` (default; sort + cumsum there) is the Euclidean projection: here the
threshold is found exactly, in registers, by a monotone Newton iteration on f(theta) = sum max(u - theta, 0), inside the
fused pass (csrc/simplex4.h, sell.h).  ``bisection_search`` is NOT the same map (simplex.py:6-123: feasible columns are
returned unclamped, the shift uses max(x / z), nu is bisected to 1e-6): it is restated as its own kernel
(``project_dense_bisect_kernel``), and inside a matching objective entries that select it take the dense-block route of
operators without a fused form (objectives/matching.py:_CustomBlocks) -- nothing is silently substituted.
"""
import torch

from dualip_amd import _hip
from dualip_amd.projections.base import ProjectionOperator, _apply_dense, register

_METHODS = ("duchi", "bisection_search")


class _SimplexBase(ProjectionOperator):
    _kind = _hip.PROJ_SIMPLEX

    def __init__(self, z: float = 1.0, method: str = "duchi"):
        self.z = z
        self.proj_method = method
        if self.proj_method not in _METHODS:
            raise ValueError(f"Unsupported projection method: {self.proj_method}")

    def descriptor(self):
        """Fused-pass form: the exact projection only (None for ``bisection_search``: dense-block route)."""
        assert self.z > 0, "Simplex radius z must be positive."
        if self.proj_method == "bisection_search":
            return None
        return _hip.ProjDesc(self._kind, 0, float(self.z), 0.0)

    def dense_descriptor(self) -> _hip.ProjDesc:
        assert self.z > 0, "Simplex radius z must be positive."
        return _hip.ProjDesc(self._kind, _hip.PROJ_FLAG_BISECTION if self.proj_method == "bisection_search" else 0, float(self.z), 0.0)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        # a vector comes back as an [L, 1] block, as the reference does (simplex.py:248-249)
        return _apply_dense(self, x, force_2d=True)


@register("simplex")
class SimplexIneq(_SimplexBase):
    _kind = _hip.PROJ_SIMPLEX


@register("simplex_eq")
class SimplexEq(_SimplexBase):
    _kind = _hip.PROJ_SIMPLEX_EQ


def _dense_in_working_precision(op: _SimplexBase, x: torch.Tensor) -> torch.Tensor:
    """``op`` on a [L, B] block whose dtype the library has no kernel for (bfloat16 / float16: the reference's tests project such blocks):
    projected in float32 and rounded back once -- never less accurate than the reference's arithmetic in the narrow type."""
    if x.dtype in (torch.float32, torch.float64):
        return _apply_dense(op, x, force_2d=True)
    return _apply_dense(op, x.to(torch.float32), force_2d=True).to(x.dtype)


def _duchi_proj(x: torch.Tensor, z: float, inequality: bool = False, tol: float = 1e-6, cols_per_chunk: int = 10000) -> torch.Tensor:
    """The reference's module-level function (simplex.py:126-236), which its own tests import: every column of the [L, B] block ``x``
    projected onto {w >= 0, sum w = z} (``inequality=True``: sum w <= z).  One ``dl_project_dense`` launch; ``tol`` is the reference's
    1e-6 feasibility slack (fixed in the kernel) and ``cols_per_chunk`` bounded the reference's temporaries -- neither changes the result."""
    assert z > 0, "Simplex radius z must be positive."
    if tol != 1e-6:
        raise ValueError("the kernel's feasibility slack is the reference's default 1e-6")
    return _dense_in_working_precision((SimplexIneq if inequality else SimplexEq)(z=z, method="duchi"), x)


def _proj_via_bisection_search(x: torch.Tensor, z: float = 1.0, inequality: bool = False, tol: float = 1e-6, max_iter: int = 50) -> torch.Tensor:
    """The reference's bisection variant (simplex.py:6-123) under its own name: ``project_dense_bisect_kernel`` through ``dl_project_dense``."""
    assert z > 0, "Simplex radius z must be positive."
    return _dense_in_working_precision((SimplexIneq if inequality else SimplexEq)(z=z, method="bisection_search"), x)
