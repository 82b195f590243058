"""CSC helpers used around the hot path (data preparation, sharding).

Reference: src/dualip/utils/sparse_utils.py.  The per-iteration primitives of that module (left_multiply_sparse,
elementwise_csc, apply_F_to_columns, row_sums_csc) have no counterpart here on purpose: their work is fused into
``dl_matching_calculate`` (csrc/matching_kernels.hip).  What remains are the one-off structural helpers.
"""
from typing import List, Sequence

import torch


def _require_csc(M: torch.Tensor, name: str = "M") -> None:
    if M.layout != torch.sparse_csc:
        raise ValueError(f"{name} must be CSC-format sparse")


def split_csc_by_cols(M: torch.Tensor, split_sizes: Sequence[int]) -> List[torch.Tensor]:
    """Cut a CSC matrix into consecutive column blocks of the given widths (reference :246-290).

    Each block owns fresh arrays with its column pointer re-based to 0.  All boundaries are read with ONE
    device-to-host transfer (the reference does two ``.item()`` synchronisations per block).
    """
    _require_csc(M)
    m, n = M.shape
    sizes = [int(s) for s in split_sizes]
    if sum(sizes) != n:
        raise ValueError(f"split_sizes must sum to {n}")
    colptr, rowidx, vals = M.ccol_indices(), M.row_indices(), M.values()
    bounds = [0]
    for s in sizes:
        bounds.append(bounds[-1] + s)
    cuts = colptr[torch.tensor(bounds, device=colptr.device)].tolist()
    blocks = []
    for i, width in enumerate(sizes):
        k0, k1 = int(cuts[i]), int(cuts[i + 1])
        sub_ptr = colptr[bounds[i] : bounds[i + 1] + 1] - k0
        blocks.append(torch.sparse_csc_tensor(sub_ptr, rowidx[k0:k1].clone(), vals[k0:k1].clone(), size=(m, width)))
    return blocks


def hstack_csc(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    """Column-wise concatenation of CSC matrices with equal row count / dtype / device (reference :293-350)."""
    first = tensors[0]
    rows, dtype, device = first.size(0), first.dtype, first.device
    for i, t in enumerate(tensors):
        _require_csc(t, f"tensor {i}")
        if t.size(0) != rows:
            raise ValueError(f"tensor {i} has {t.size(0)} rows, expected {rows}")
        if t.dtype != dtype:
            raise TypeError("all tensors must share the same dtype")
        if t.device != device:
            raise TypeError("all tensors must be on the same device")
    ptrs, offset = [first.ccol_indices()[:1]], 0
    for t in tensors:
        ptrs.append(t.ccol_indices()[1:] + offset)
        offset += int(t.values().shape[0])
    return torch.sparse_csc_tensor(
        torch.cat(ptrs),
        torch.cat([t.row_indices() for t in tensors]),
        torch.cat([t.values() for t in tensors]),
        size=(rows, sum(int(t.size(1)) for t in tensors)),
    )


def row_norms_csc(A: torch.Tensor) -> torch.Tensor:
    """L2 norm of every row of a CSC matrix (reference :429-450); device tensors go through the HIP scatter."""
    _require_csc(A, "A")
    from dualip_amd.preprocessing.precondition import _row_norms  # local import: avoids a cycle

    return _row_norms(A)
