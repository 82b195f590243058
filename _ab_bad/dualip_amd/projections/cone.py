"""One-sided bound: the registry's ``"cone"`` operator (reference: src/dualip/projections/cone.py:6-28).

``lower`` gives [lower, +inf), ``upper`` gives (-inf, upper], neither gives the identity and both is rejected with the
reference's message.  Kernel side these are the DL_PROJ_CONE_LOWER / CONE_UPPER / NONE kinds: the same clamp as the box
with an infinite bound on the open side.
"""
import math

from dualip_amd import _hip
from dualip_amd.projections.base import ProjectionOperator, register


@register("cone")
class coneProjection(ProjectionOperator):  # (lower-case initial as in the reference: the name is part of the API)
    def __init__(self, lower=None, upper=None):
        if lower is not None and upper is not None:
            raise ValueError("Only one of 'lower' or 'upper' should be specified, not both.")
        self.lower = lower
        self.upper = upper

    def bounds(self):
        """(lower, upper) with infinities on the open side(s)."""
        return (-math.inf if self.lower is None else float(self.lower), math.inf if self.upper is None else float(self.upper))

    def descriptor(self) -> _hip.ProjDesc:
        if self.lower is not None:
            return _hip.ProjDesc(_hip.PROJ_CONE_LOWER, 0, float(self.lower), 0.0)
        if self.upper is not None:
            return _hip.ProjDesc(_hip.PROJ_CONE_UPPER, 0, float(self.upper), 0.0)
        return _hip.ProjDesc(_hip.PROJ_NONE, 0, 0.0, 0.0)

    def __repr__(self) -> str:
        return f"coneProjection(lower={self.lower}, upper={self.upper})"
