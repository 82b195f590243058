"""Box projection (reference: src/dualip/projections/box.py:6-16)."""
from dualip_amd import _hip
from dualip_amd.projections.base import ProjectionOperator, register


@register("box")
class BoxProjection(ProjectionOperator):
    """Coordinate-wise clamp to [lower, upper] (defaults 0 / 1)."""

    def __init__(self, lower: float = 0.0, upper: float = 1.0):
        self.lower, self.upper = lower, upper

    def descriptor(self) -> _hip.ProjDesc:
        return _hip.ProjDesc(_hip.PROJ_BOX, 0, float(self.lower), float(self.upper))
