"""Projection registry; importing the package registers the built-in operators (box, cone, simplex, simplex_eq)."""
from dualip_amd.projections import box, cone, simplex  # noqa: F401  (registration side effect)
from dualip_amd.projections.base import ProjectionEntry, ProjectionOperator, create_projection_map, project, register

__all__ = ["project", "register", "ProjectionOperator", "create_projection_map", "ProjectionEntry"]
