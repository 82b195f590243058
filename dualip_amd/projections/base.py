"""Projection-map API: ProjectionEntry records, the operator registry and the map builder.

Reference: src/dualip/projections/base.py:8-97 (names, key format and error behaviour are kept).  Every operator
additionally exposes ``descriptor()`` -- the (kind, p0, p1) record the fused HIP kernel consumes -- and applies itself
to dense blocks through ``dl_project_dense`` (include/dualip_hip.h), i.e. on the GPU, not with ATen ops.
"""
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from typing import Optional, Dict, List, Sequence, Union

import torch

from dualip_amd import _hip

# contiguous index sets larger than this stay ``range`` objects (a 100M-entity map must not become a Python list)
_LIST_LIMIT = 1 << 20


@dataclass
class ProjectionEntry:
    proj_type: str = ""
    proj_params: dict = field(default_factory=dict)
    # list[int] in the reference; a ``range`` or an integer tensor is accepted as well (and preferred at scale)
    indices: Union[List[int], range, torch.Tensor] = field(default_factory=list)


class ProjectionOperator(ABC):
    """Callable projection onto a convex set; ``__call__`` never modifies its input."""

    @abstractmethod
    def __init__(self, **params):
        ...

    def descriptor(self) -> Optional[_hip.ProjDesc]:
        """Kernel-side description of this operator, or None: an operator the fused kernel does not know (a user's own
        subclass that only defines ``__call__``, as the reference's interface asks) is applied to zero-padded dense
        blocks of columns instead, like the reference's apply_F_to_columns does for every operator."""
        return None

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self.descriptor() is None:
            raise NotImplementedError("a ProjectionOperator defines descriptor() (built-in kinds) or overrides __call__")
        return _apply_dense(self, x)


def _apply_dense(op: "ProjectionOperator", x: torch.Tensor, force_2d: bool = False) -> torch.Tensor:
    if not x.is_cuda:  # a CPU vector / block (the reference's tests): projected on the current ROCm device, answered on the CPU
        return _apply_dense(op, _hip.stage(x, "projection input"), force_2d).to(x.device)
    lib = _hip.load()
    squeeze = x.ndim == 1
    x2 = x.unsqueeze(1) if squeeze else x
    if x2.ndim != 2:
        raise ValueError("projection operators take a vector [L] or a block [L, K] with one vector per column")
    src = x2.contiguous()
    out = torch.empty_like(src)
    desc = op.dense_descriptor() if hasattr(op, "dense_descriptor") else op.descriptor()
    with torch.cuda.device(x.device):
        rc = lib.dl_project_dense(
            src.shape[0], src.shape[1], _hip.dtype_code(src.dtype), _hip.ptr(src), _hip.ptr(out), desc, _hip.stream_ptr(x.device)
        )
    _hip.check(rc)
    return out.squeeze(1) if (squeeze and not force_2d) else out


_registry: Dict[str, type] = {}


def register(name):
    """Class decorator: make an operator constructible through ``project(name, **params)``."""

    def deco(cls):
        _registry[name] = cls
        return cls

    return deco


def project(name: str, **params) -> ProjectionOperator:
    try:
        cls = _registry[name]
    except KeyError:
        raise ValueError(f"Unknown projection operator '{name}'") from None
    return cls(**params)


def create_projection_map(
    proj_type: str,
    proj_params: Dict[str, float],
    num_indices: int,
    indices: Union[Sequence[int], torch.Tensor, None] = None,
    key_prefix: str = "",
) -> Dict[str, ProjectionEntry]:
    """One-entry projection map ``{key: ProjectionEntry}``; key = ``{prefix}{type}_{k}_{v}...`` with the parameter
    names sorted (e.g. ``simplex_z_1.0``, ``box_lower_0.0_upper_1.0``)."""
    if indices is None:
        indices = list(range(num_indices)) if num_indices <= _LIST_LIMIT else range(num_indices)
    tail = "_".join(f"{k}_{v}" for k, v in sorted(proj_params.items()))
    key = f"{key_prefix}{proj_type}_{tail}"
    return {key: ProjectionEntry(proj_type=proj_type, proj_params=proj_params, indices=indices)}
