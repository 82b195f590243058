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


# Relative cost of one non-zero in the fused pass, by projection operator (measured on MI355X, 100M entities x 10k destinations,
# fp32, fused-kernel ms per launch: all-box 1.46-1.50; all-simplex 1.65-1.69 early in a solve, 1.71-1.73 late).  Point-wise
# operators (box, cone, identity) stream at the read ceiling; the simplex operators pay for their Newton passes.
PROJECTION_COST = {"simplex": 1.14, "simplex_eq": 1.14}


def projection_cost(proj_type: str) -> float:
    return PROJECTION_COST.get(proj_type, 1.0)


def contiguous_cuts(num_cols: int, num_parts: int, blocks: Sequence[Tuple[int, int, float]] = (), align: int = 1) -> List[int]:
    """``num_parts + 1`` cut points ``0 = t_0 <= t_1 <= ... <= t_W = num_cols``: shard r holds the CONTIGUOUS columns
    [t_r, t_{r+1}) -- the shape of the reference's split (dist_utils.py:53-57), so a shard's projection map is a re-base of
    the global one.

    ``blocks`` empty: the reference's sizes, n // W (+1 for the first n % W shards) -- count balanced.
    ``blocks`` = (lo, hi, weight) ranges of columns with a relative cost per column (columns outside every block weigh 1):
    the cuts equalise the shards' total COST instead of their column counts.  For a map made of a few large blocks with
    different operators (box on one half of the entities, simplex on the other) the count-balanced split gives some ranks
    only the cheap operator and others only the expensive one (measured at 100M entities: 1.46 ms against 1.70 ms per
    iteration), and every iteration waits for the slowest rank.
    ``align`` rounds interior cuts to multiples of a chunk size (the synthetic generator's chunks)."""
    if num_parts < 1:
        raise ValueError("num_parts must be positive")
    if not blocks:
        cuts, pos = [0], 0
        if align > 1:  # count-balanced in units of `align` columns
            units = -(-num_cols // align)
            for s in balanced_split_sizes(units, num_parts):
                pos = min(num_cols, pos + s * align)
                cuts.append(pos)
        else:
            for s in balanced_split_sizes(num_cols, num_parts):
                pos += s
                cuts.append(pos)
        cuts[-1] = num_cols
        return cuts
    # piecewise-constant weight over [0, num_cols)
    pieces, pos = [], 0
    for lo, hi, wgt in sorted((int(a), int(b), float(w)) for a, b, w in blocks if int(b) > int(a)):
        if lo < pos:
            raise ValueError("cost blocks must not overlap")
        if lo > pos:
            pieces.append((pos, lo, 1.0))
        pieces.append((lo, min(hi, num_cols), wgt))
        pos = min(hi, num_cols)
    if pos < num_cols:
        pieces.append((pos, num_cols, 1.0))
    total = sum((hi - lo) * w for lo, hi, w in pieces)
    cuts = [0]
    for k in range(1, num_parts):
        target, acc, t = total * k / num_parts, 0.0, num_cols
        for lo, hi, w in pieces:
            cost = (hi - lo) * w
            if acc + cost >= target and w > 0:
                t = lo + (target - acc) / w
                break
            acc += cost
        t = int(round(t / align)) * align if align > 1 else int(round(t))
        cuts.append(min(num_cols, max(cuts[-1], t)))
    cuts.append(num_cols)
    return cuts


def shard_costs(cuts: Sequence[int], blocks: Sequence[Tuple[int, int, float]]) -> List[float]:
    """Estimated cost of every shard of ``cuts`` under the block weights (columns outside every block weigh 1)."""
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        cost = float(b - a)
        for lo, hi, w in blocks:
            cost += max(0, min(b, hi) - max(a, lo)) * (float(w) - 1.0)
        out.append(cost)
    return out


def projection_cost_blocks(projection_map: Dict[str, ProjectionEntry]) -> List[Tuple[int, int, float]]:
    """(lo, hi, weight) for every entry of a projection map whose indices are one contiguous ``range`` and whose operator
    does not cost 1 -- the ``blocks`` argument of ``contiguous_cuts``.  Entries with scattered index lists are left at
    weight 1 (a cost-weighted contiguous cut has nothing to gain from them)."""
    out = []
    for entry in projection_map.values():
        idx = entry.indices
        w = projection_cost(entry.proj_type)
        if isinstance(idx, range) and idx.step == 1 and len(idx) and w != 1.0:
            out.append((idx.start, idx.stop, w))
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
