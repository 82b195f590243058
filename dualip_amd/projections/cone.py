"""One-sided cone projection (reference: src/dualip/projections/cone.py:6-28)."""
from dualip_amd import _hip
from dualip_amd.projections.base import ProjectionOperator, register


@register("cone")
class coneProjection(ProjectionOperator):
    """[lower, +inf) or (-inf, upper] per coordinate; identity when neither bound is given; both is an error."""

    def __init__(self, lower=None, upper=None):
        if lower is not None and upper is not None:
            raise ValueError("Only one of 'lower' or 'upper' should be specified, not both.")
        self.lower, self.upper = lower, upper

    def descriptor(self) -> _hip.ProjDesc:
        if self.lower is not None:
            return _hip.ProjDesc(_hip.PROJ_CONE_LOWER, 0, float(self.lower), 0.0)
        if self.upper is not None:
            return _hip.ProjDesc(_hip.PROJ_CONE_UPPER, 0, float(self.upper), 0.0)
        return _hip.ProjDesc(_hip.PROJ_NONE, 0, 0.0, 0.0)
