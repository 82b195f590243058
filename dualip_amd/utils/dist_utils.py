"""Column sharding helpers (reference: src/dualip/utils/dist_utils.py:9-71).

Entities (columns) are independent given the dual vector, so a problem is sharded by contiguous column ranges of
sizes n//W (+1 for the first n%W ranks).  Index sets are kept as ``range`` objects whenever they are contiguous, so
sharding a 100M-entity problem does not materialise Python lists.
"""
from typing import Dict, List, Sequence, Tuple, Union

import torch

from dualip_amd.projections.base import _LIST_LIMIT, ProjectionEntry
from dualip_amd.utils.sparse_utils import split_csc_by_cols


def balanced_split_sizes(num_cols: int, num_parts: int) -> List[int]:
    base, extra = divmod(num_cols, num_parts)
    return [base + (1 if i < extra else 0) for i in range(num_parts)]


def balanced_block_ranges(blocks: Sequence[Tuple[int, int]], num_parts: int, part: int, align: int = 1) -> List[Tuple[int, int]]:
    """Column ranges of shard ``part`` when every block ``(lo, hi)`` of columns is split evenly over the shards.

    A projection map made of a few large blocks with different operators (say box on one half of the entities and simplex
    on the other) is badly served by the reference's contiguous split (dist_utils.py:49-62): some ranks would hold only the
    cheap operator, others only the expensive one, and every iteration waits for the slowest.  Giving each rank its share of
    EVERY block equalises the work; columns are independent given the dual vector, so any partition yields the same sums.
    ``align`` keeps the cut points on multiples of a chunk size (relative to each block's start)."""
    out = []
    for lo, hi in blocks:
        units = -(-(hi - lo) // align)
        sizes = balanced_split_sizes(units, num_parts)
        u0 = sum(sizes[:part])
        a = min(hi, lo + u0 * align)
        b = min(hi, a + sizes[part] * align)
        if b > a:
            out.append((a, b))
    return out


def _as_plain(indices):
    if isinstance(indices, range) and len(indices) <= _LIST_LIMIT:
        return list(indices)
    return indices


def _localise(indices, local_cols):
    """Positions (in ``local_cols``) of the members of ``indices`` that belong to this shard, in ``indices`` order."""
    contiguous = isinstance(local_cols, range) and local_cols.step == 1
    if contiguous and isinstance(indices, range) and indices.step == 1:
        lo, hi = max(indices.start, local_cols.start), min(indices.stop, local_cols.stop)
        return _as_plain(range(lo - local_cols.start, max(lo, hi) - local_cols.start))
    if contiguous:
        idx = torch.as_tensor(indices, dtype=torch.int64)
        keep = idx[(idx >= local_cols.start) & (idx < local_cols.stop)] - local_cols.start
        return keep.tolist()
    lookup = {g: loc for loc, g in enumerate(local_cols)}
    return [lookup[g] for g in (indices.tolist() if isinstance(indices, torch.Tensor) else indices) if g in lookup]


def global_to_local_projection_map(global_map: Dict[str, ProjectionEntry], local_cols: Union[Sequence[int], range]) -> Dict[str, ProjectionEntry]:
    """Re-base a global projection map to the columns held by one shard; keys without local columns are dropped."""
    local_map: Dict[str, ProjectionEntry] = {}
    for key, entry in global_map.items():
        local = _localise(entry.indices, local_cols)
        if len(local):
            local_map[key] = ProjectionEntry(proj_type=entry.proj_type, proj_params=entry.proj_params, indices=local)
    return local_map


def split_tensors_to_devices(a_mat: torch.Tensor, c_mat: torch.Tensor, compute_devices: list) -> Tuple[list, list, list]:
    """Split A and c by columns, one block per device.  Returns (A_blocks, c_blocks, split_index_map) where
    split_index_map[i] is the ``range`` of global columns of block i (a flat index list for the empty-device case,
    as in the reference)."""
    if a_mat.layout != torch.sparse_csc or c_mat.layout != torch.sparse_csc:
        raise ValueError("Both A and B must be CSC-format sparse tensors")
    n = a_mat.size(1)
    if not compute_devices:
        return [a_mat], [c_mat], list(range(n))
    sizes = balanced_split_sizes(n, len(compute_devices))
    index_map, start = [], 0
    for s in sizes:
        index_map.append(range(start, start + s))
        start += s
    a_blocks = [blk.to(dev) for blk, dev in zip(split_csc_by_cols(a_mat, sizes), compute_devices)]
    c_blocks = [blk.to(dev) for blk, dev in zip(split_csc_by_cols(c_mat, sizes), compute_devices)]
    return a_blocks, c_blocks, index_map
