"""Box projection: the registry's ``"box"`` operator (reference: src/dualip/projections/box.py:6-16).

On the device a box is one v_med3_f32 inside the fused pass (csrc/simplex.h: clamp3); this class only carries the two
bounds to the kernel-side descriptor and applies itself to dense blocks through dl_project_dense.
"""
from dualip_amd import _hip
from dualip_amd.projections.base import ProjectionOperator, register


@register("box")
class BoxProjection(ProjectionOperator):
    """x -> min(max(x, lower), upper) coordinate by coordinate; the unit box when no bound is given."""

    def __init__(self, lower: float = 0.0, upper: float = 1.0):
        self.lower = lower
        self.upper = upper

    def bounds(self):
        """(lower, upper) as floats -- what the generic-LP objective turns into per-variable clamp arrays."""
        return float(self.lower), float(self.upper)

    def descriptor(self) -> _hip.ProjDesc:
        lo, hi = self.bounds()
        return _hip.ProjDesc(_hip.PROJ_BOX, 0, lo, hi)

    def __repr__(self) -> str:
        return f"BoxProjection(lower={self.lower}, upper={self.upper})"
