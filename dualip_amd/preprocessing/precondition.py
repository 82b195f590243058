"""Jacobi (row-norm) pre-conditioning on the device.

Reference: src/dualip/preprocessing/precondition.py:8-60.  ``jacobi_precondition`` scales every row of A and b in
place by 1/||A_i||_2 with one scatter pass + one scale pass (``dl_jacobi_precondition``); the inverse maps a dual
vector of the scaled problem back.
"""
from pathlib import Path
from typing import Union

import torch

from dualip_amd import _hip


def _jacobi_call(A: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    vals = A.values()
    if not vals.is_cuda or not b.is_cuda:  # CPU-resident A / b: scaled on the current ROCm device, written back IN PLACE (the reference's contract)
        dev = vals.device if vals.is_cuda else (b.device if b.is_cuda else _hip.compute_device())
        A_dev, b_dev = _hip.stage(A, "A", dev) if not vals.is_cuda else A, _hip.stage(b, "b", dev) if not b.is_cuda else b
        norms = _jacobi_call(A_dev, b_dev)
        if not vals.is_cuda:
            vals.copy_(A_dev.values())
        if not b.is_cuda:
            b.copy_(b_dev)
        return norms.to(vals.device)
    rowidx = A.row_indices().contiguous()
    norms = torch.empty(A.size(0), dtype=vals.dtype, device=vals.device)
    with torch.cuda.device(vals.device):
        rc = _hip.load().dl_jacobi_precondition(
            A.size(0), vals.shape[0], _hip.ptr(rowidx), _hip.idx_code(rowidx.dtype), _hip.ptr(vals), _hip.ptr(b), _hip.ptr(norms),
            _hip.dtype_code(vals.dtype), _hip.stream_ptr(vals.device),
        )
    _hip.check(rc)
    return norms


def _row_norms(A: torch.Tensor) -> torch.Tensor:
    scratch = torch.sparse_csc_tensor(A.ccol_indices(), A.row_indices(), A.values().clone(), size=A.shape)
    return _jacobi_call(scratch, torch.ones(A.size(0), dtype=A.values().dtype, device=A.values().device))


def jacobi_precondition(A: torch.Tensor, b: torch.Tensor, norms_save_path: str = None) -> torch.Tensor:
    """Scale A (CSC values) and b IN PLACE by the reciprocal row L2 norms; returns the row norms."""
    if A.layout != torch.sparse_csc:
        raise ValueError("Expected M to be a CSC-format sparse tensor")
    if not (A.values().is_contiguous() and b.is_contiguous()):
        raise ValueError("A.values() and b must be contiguous to be scaled in place")
    norms = _jacobi_call(A, b)
    if norms_save_path:
        torch.save(norms, Path(norms_save_path))
    return norms


def jacobi_invert_precondition(dual_val: torch.Tensor, norms_path_or_tensor: Union[str, torch.Tensor]) -> torch.Tensor:
    """lambda_original = lambda_scaled / row_norms  (scaling Ax - b by D scales lambda by D^-1)."""
    if isinstance(norms_path_or_tensor, str):
        norms = torch.load(Path(norms_path_or_tensor), map_location=dual_val.device)
    else:
        norms = norms_path_or_tensor.to(dual_val.device)
    return dual_val / norms
