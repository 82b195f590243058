"""CSC helpers (reference: src/dualip/utils/sparse_utils.py).

Inside a solve the per-iteration primitives of that module -- left_multiply_sparse, elementwise_csc, apply_F_to_columns,
row_sums_csc -- do not run as separate operations here: their work is fused into ``dl_matching_calculate``
(csrc/matching_kernels4.hip).  They are still offered stand-alone, with the reference's signatures, each as ONE HIP launch
(csrc/csc_ops.hip).  There is no CPU implementation: CPU tensors (the reference's tests, ``host_device="cpu"`` callers) are copied to
the current ROCm device, the launch runs there, and the result -- or the ``output_tensor`` written in place -- comes back on the CPU
(``_cpu_callers`` below; ``HipLibraryError`` when the process sees no GPU).  Callables the
library has no kernel for (an arbitrary ``op`` / ``F_batch``) are applied with the caller's own torch code on the device.
"""
import ctypes
import functools
import inspect
import operator
from typing import Callable, List, Optional, Sequence

import torch

from dualip_amd import _hip


def _cpu_callers(fn):
    """CPU tensors among the arguments: every tensor argument is copied to one ROCm device (that of a device argument if there is one, else
    the current device), ``fn`` runs there, and the caller gets what the reference's CPU code would have given it: a CPU result, and an
    ``output_tensor`` whose values were overwritten in place (the return value is then that tensor's values, as in the reference)."""
    sig = inspect.signature(fn)

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        bound = sig.bind(*args, **kwargs)
        tens = [v for v in bound.arguments.values() if isinstance(v, torch.Tensor)]
        if not tens or all(t.is_cuda for t in tens):
            return fn(*args, **kwargs)
        dev = next((t.device for t in tens if t.is_cuda), None) or _hip.compute_device()
        out_cpu = bound.arguments.get("output_tensor")
        home = next(t.device for t in tens if not t.is_cuda)
        for k, v in list(bound.arguments.items()):
            if isinstance(v, torch.Tensor) and not v.is_cuda:
                bound.arguments[k] = _hip.stage(v, f"{fn.__name__}({k})", dev)
        res = fn(*bound.args, **bound.kwargs)
        if isinstance(out_cpu, torch.Tensor) and not out_cpu.is_cuda:
            out_cpu.values().copy_(bound.arguments["output_tensor"].values())
            return out_cpu.values()
        return res.to(home) if isinstance(res, torch.Tensor) else res

    return wrapper


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


# ---------------------------------------------------------------------------------------------------------
# stand-alone per-iteration primitives (reference :7-243)
# ---------------------------------------------------------------------------------------------------------
def _csc_parts(M: torch.Tensor, name: str = "M"):
    vals = M.values()
    _hip.require_device(vals, name)  # (CPU callers were staged by _cpu_callers before this point)
    if not vals.is_contiguous():
        raise ValueError("CSC value arrays must be contiguous")
    return M.ccol_indices().contiguous(), M.row_indices().contiguous(), vals


def _finish(M: torch.Tensor, new_vals: torch.Tensor, output_tensor: Optional[torch.Tensor]):
    """The reference's return convention: a new CSC tensor with M's pattern, or the output tensor's values after copy_."""
    if output_tensor is None:
        return torch.sparse_csc_tensor(M.ccol_indices(), M.row_indices(), new_vals, size=M.size())
    out_vals = output_tensor.values()
    if out_vals.data_ptr() != new_vals.data_ptr():
        out_vals.copy_(new_vals)
    return out_vals


def _target(vals: torch.Tensor, output_tensor: Optional[torch.Tensor]) -> torch.Tensor:
    """Where a kernel writes: straight into the output tensor's values when they can take it, else a fresh array."""
    if output_tensor is not None:
        ov = output_tensor.values()
        if ov.is_contiguous() and ov.dtype == vals.dtype and ov.shape == vals.shape and ov.device == vals.device:
            return ov
    return torch.empty_like(vals)


def dot_product_csc(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """sum_ij A_ij B_ij of two CSC tensors with the same pattern (reference :7-23)."""
    _require_csc(A, "A")
    _require_csc(B, "B")
    if A.shape != B.shape:
        raise AssertionError(f"Expected shapes (m, n) and (m, n), got {A.shape} and {B.shape}")
    return torch.dot(A.values(), B.values())


_OPS = {torch.add: 0, operator.add: 0, torch.sub: 1, torch.subtract: 1, operator.sub: 1, torch.mul: 2, torch.multiply: 2, operator.mul: 2,
        torch.div: 3, torch.divide: 3, torch.true_divide: 3, operator.truediv: 3}


@_cpu_callers
def elementwise_csc(A: torch.Tensor, B: torch.Tensor, op, output_tensor: Optional[torch.Tensor] = None):
    """``op`` applied to the values of two CSC tensors with identical pattern (reference :26-51).  torch / operator add, sub, mul
    and div run as one HIP launch; any other callable is applied to the two value tensors on the device."""
    if A.layout != torch.sparse_csc or B.layout != torch.sparse_csc:
        raise ValueError("Both A and B must be CSC-format sparse tensors")
    if output_tensor is None and not (torch.equal(A.ccol_indices(), B.ccol_indices()) and torch.equal(A.row_indices(), B.row_indices())):
        raise ValueError("A and B must share the same sparsity pattern")
    _, _, va = _csc_parts(A, "A")
    _, _, vb = _csc_parts(B, "B")
    code = _OPS.get(op)
    if code is None or va.dtype != vb.dtype or va.dtype not in (torch.float32, torch.float64):
        return _finish(A, op(va, vb), output_tensor)
    out = _target(va, output_tensor)
    with torch.cuda.device(va.device):
        _hip.check(_hip.load().dl_csc_elementwise(va.numel(), _hip.ptr(va), _hip.ptr(vb), _hip.ptr(out), code, _hip.dtype_code(va.dtype), _hip.stream_ptr(va.device)))
    return _finish(A, out, output_tensor)


@_cpu_callers
def left_multiply_sparse(v: torch.Tensor, M: torch.Tensor, output_tensor: Optional[torch.Tensor] = None):
    """diag(v) @ M for a CSC matrix, pattern preserved (reference :54-85)."""
    if M.layout != torch.sparse_csc:
        raise ValueError("Expected M to be a CSC-format sparse tensor")
    _, rowidx, vals = _csc_parts(M)
    _hip.require_device(v, "v")
    v = v.to(vals.dtype).contiguous()
    out = _target(vals, output_tensor)
    with torch.cuda.device(vals.device):
        _hip.check(_hip.load().dl_csc_scale_rows(vals.numel(), _hip.ptr(rowidx), _hip.idx_code(rowidx.dtype), _hip.ptr(vals), _hip.ptr(v), _hip.ptr(out),
                                                 _hip.dtype_code(vals.dtype), _hip.stream_ptr(vals.device)))
    return _finish(M, out, output_tensor)


@_cpu_callers
def right_multiply_sparse(M: torch.Tensor, v: torch.Tensor, output_tensor: Optional[torch.Tensor] = None):
    """M @ diag(v) for a CSC matrix, pattern preserved (reference :88-130; no per-column host loop)."""
    if M.layout != torch.sparse_csc:
        raise ValueError("Expected M to be a CSC-format sparse tensor")
    colptr, _, vals = _csc_parts(M)
    _hip.require_device(v, "v")
    v = v.to(vals.dtype).contiguous()
    out = _target(vals, output_tensor)
    with torch.cuda.device(vals.device):
        _hip.check(_hip.load().dl_csc_scale_cols(int(M.size(1)), vals.numel(), _hip.ptr(colptr), _hip.idx_code(colptr.dtype), _hip.ptr(vals), _hip.ptr(v), _hip.ptr(out),
                                                 _hip.dtype_code(vals.dtype), _hip.stream_ptr(vals.device)))
    return _finish(M, out, output_tensor)


@_cpu_callers
def row_sums_csc(A: torch.Tensor) -> torch.Tensor:
    """Dense vector of the row sums of a CSC matrix (reference :223-243), accumulated in float64 and rounded once."""
    _require_csc(A, "A")
    _, rowidx, vals = _csc_parts(A, "A")
    out = torch.empty(int(A.size(0)), dtype=vals.dtype, device=vals.device)
    with torch.cuda.device(vals.device):
        _hip.check(_hip.load().dl_csc_row_sums(int(A.size(0)), vals.numel(), _hip.ptr(rowidx), _hip.idx_code(rowidx.dtype), _hip.ptr(vals), _hip.ptr(out),
                                               _hip.dtype_code(vals.dtype), _hip.stream_ptr(vals.device)))
    return out


@_cpu_callers
def apply_F_to_columns(M: torch.Tensor, F_batch: Callable[[torch.Tensor], torch.Tensor], buckets: Sequence[torch.Tensor], output_tensor: Optional[torch.Tensor] = None):
    """Replace the values of every column listed in ``buckets`` by ``F_batch`` of them (reference :133-220).

    A projection operator of this package with a kernel form (box, cone, simplex, simplex_eq -- what ``project(...)`` returns)
    is applied per column in ONE launch per bucket, over the column's own entries.  Any other callable takes the reference's
    route on the device: a zero-padded [L x K] block per bucket, ``F_batch(block)``, valid entries scattered back.
    Entries of columns in no bucket keep their input value (the reference leaves them uninitialised, :177)."""
    assert M.layout == torch.sparse_csc, "M must be a CSC sparse tensor"
    colptr, _, vals = _csc_parts(M)
    device = vals.device
    desc = F_batch.descriptor() if hasattr(F_batch, "descriptor") else None
    new_vals = vals.clone()
    lib = _hip.load()
    for cols in buckets:
        cols = torch.as_tensor(cols, dtype=torch.int64, device=device).contiguous()
        K = int(cols.numel())
        if K == 0:
            continue
        if desc is not None and vals.dtype in (torch.float32, torch.float64):
            with torch.cuda.device(device):
                _hip.check(lib.dl_csc_project_columns(K, _hip.ptr(cols), _hip.ptr(colptr), _hip.idx_code(colptr.dtype), _hip.ptr(vals), _hip.ptr(new_vals),
                                                      ctypes.byref(desc), _hip.dtype_code(vals.dtype), _hip.stream_ptr(device)))
            continue
        starts = colptr[cols].to(torch.int64)
        lengths = colptr[cols + 1].to(torch.int64) - starts
        total = int(lengths.sum())
        if total == 0:
            continue
        L = int(lengths.max())
        cols_rep = torch.arange(K, device=device).repeat_interleave(lengths)
        idx_in_col = torch.arange(total, device=device) - (lengths.cumsum(0) - lengths)[cols_rep]
        flat = starts[cols_rep] + idx_in_col
        block = torch.zeros((L, K), device=device, dtype=vals.dtype)
        block[idx_in_col, cols_rep] = vals[flat]
        new_vals[flat] = F_batch(block)[idx_in_col, cols_rep]
    return _finish(M, new_vals, output_tensor)


def vstack_csc(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    """Row-wise concatenation of CSC matrices with equal column count / dtype / device (reference :351-426), without the
    per-column host loop: the blocks' entries are merged by one stable sort on the column index."""
    if not tensors:
        raise ValueError("Cannot stack empty list of tensors")
    first = tensors[0]
    n_cols, dtype, device = first.size(1), first.dtype, first.device
    for i, t in enumerate(tensors):
        _require_csc(t, f"tensor {i}")
        if t.size(1) != n_cols:
            raise ValueError(f"tensor {i} has {t.size(1)} columns, expected {n_cols}")
        if t.dtype != dtype:
            raise TypeError("all tensors must share the same dtype")
        if t.device != device:
            raise TypeError("all tensors must be on the same device")
    col_parts, row_parts, val_parts, offset = [], [], [], 0
    counts = torch.zeros(n_cols, dtype=torch.int64, device=device)
    for t in tensors:
        lens = (t.ccol_indices()[1:] - t.ccol_indices()[:-1]).to(torch.int64)
        col_parts.append(torch.repeat_interleave(torch.arange(n_cols, device=device), lens))
        row_parts.append(t.row_indices().to(torch.int64) + offset)
        val_parts.append(t.values())
        counts += lens
        offset += int(t.size(0))
    cols = torch.cat(col_parts)
    order = torch.sort(cols, stable=True).indices  # column-major; within a column block order = row order
    ccol = torch.zeros(n_cols + 1, dtype=torch.int64, device=device)
    ccol[1:] = torch.cumsum(counts, 0)
    return torch.sparse_csc_tensor(ccol, torch.cat(row_parts)[order], torch.cat(val_parts)[order], size=(offset, n_cols))
