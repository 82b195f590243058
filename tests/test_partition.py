"""How entities are split over the ranks (CPU): the reference's contiguous count-balanced cuts (dist_utils.py:53-57), the
cost-weighted contiguous cuts, the interleaved block-balanced split -- and the generator slicing that lets any cut point be
a shard boundary of the SAME global problem."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dualip_amd.utils.dist_utils import balanced_split_sizes, contiguous_cuts, projection_cost_blocks, shard_costs  # noqa: E402


def test_count_balanced_cuts_are_the_reference_sizes():
    for n, w in ((10, 3), (7, 7), (5, 8), (100_000_000, 8), (1, 1)):
        cuts = contiguous_cuts(n, w)
        assert cuts[0] == 0 and cuts[-1] == n and len(cuts) == w + 1
        assert [b - a for a, b in zip(cuts[:-1], cuts[1:])] == balanced_split_sizes(n, w)  # n // W, +1 for the first n % W


def test_cost_weighted_cuts_equalise_cost_and_stay_contiguous():
    n = 100_000_000
    blocks = [(n // 2, n, 1.14)]  # second half of the entities is 14 % dearer per column
    for w in (2, 4, 8, 3):
        cuts = contiguous_cuts(n, w, blocks)
        assert cuts[0] == 0 and cuts[-1] == n and all(b >= a for a, b in zip(cuts[:-1], cuts[1:]))
        costs = shard_costs(cuts, blocks)
        assert max(costs) / (sum(costs) / w) < 1.0 + 1e-6
        ref = shard_costs(contiguous_cuts(n, w), blocks)
        if w % 2 == 0:
            assert max(ref) / (sum(ref) / w) > 1.06  # what the count-balanced split would cost
    # a map without weights gives the count-balanced cuts up to rounding
    assert contiguous_cuts(1000, 4, [(0, 1000, 1.0)]) == contiguous_cuts(1000, 4)
    with pytest.raises(ValueError):
        contiguous_cuts(10, 2, [(0, 6, 2.0), (4, 10, 1.0)])


def test_projection_cost_blocks_reads_a_map():
    from dualip_amd.projections import create_projection_map

    pm = {**create_projection_map("box", {"lower": 0.0, "upper": 1.0}, None, indices=range(0, 60)), **create_projection_map("simplex", {"z": 1.0}, None, indices=range(60, 100))}
    assert projection_cost_blocks(pm) == [(60, 100, 1.14)]


@pytest.mark.parametrize("partition", ["contiguous", "reference", "balanced"])
@pytest.mark.parametrize("world", [1, 2, 8])
def test_bench_shard_plans_cover_every_entity_once(partition, world):
    import bench
    from benchmark.synthetic import CHUNK_COLS

    n = 6_000_000
    seen = np.zeros(n // 1000, dtype=np.int64)  # (in units of 1000 entities)
    for kind in ("mixed", "simplex"):
        seen[:] = 0
        for rank in range(world):
            ranges, pm = bench.shard_plan(kind, n, world, rank, CHUNK_COLS, partition)
            width = sum(b - a for a, b in ranges)
            assert sum(len(e.indices) for e in pm.values()) == width  # the local map covers the local columns, re-based
            pos = 0
            for e in pm.values():
                assert e.indices.start == pos
                pos = e.indices.stop
            if partition != "balanced" and ranges:
                assert all(r1[0] == r0[1] for r0, r1 in zip(ranges[:-1], ranges[1:]))  # ONE contiguous range (cut at block boundaries)
            for a, b in ranges:
                assert a % 1000 == 0 or partition == "contiguous"
                seen[a // 1000 : -(-b // 1000)] += 1
        if partition != "contiguous":
            assert (seen == 1).all()
        else:  # cuts need not be multiples of 1000: neighbours share at most the unit a cut falls into
            assert seen.min() >= 1 and (seen > 1).sum() <= world


def test_generator_shards_of_any_cut_are_slices_of_the_same_problem():
    from benchmark.synthetic import generate_matching_problem

    n, m = 600_000, 400
    full = generate_matching_problem(n, m, 0.02, device="cpu")
    cuts = [0, 100_001, 250_000, 333_333, n]  # inside a chunk, on a chunk boundary, inside the next
    parts = [generate_matching_problem(n, m, 0.02, device="cpu", col_range=(a, b)) for a, b in zip(cuts[:-1], cuts[1:])]
    A = full["input_args"].A
    assert torch.equal(A.values(), torch.cat([p["input_args"].A.values() for p in parts]))
    assert torch.equal(full["input_args"].c.values(), torch.cat([p["input_args"].c.values() for p in parts]))
    assert torch.equal(A.row_indices(), torch.cat([p["input_args"].A.row_indices() for p in parts]))
    lens = torch.cat([p["input_args"].A.ccol_indices()[1:] - p["input_args"].A.ccol_indices()[:-1] for p in parts])
    assert torch.equal(lens, A.ccol_indices()[1:] - A.ccol_indices()[:-1])
    assert torch.allclose(sum(p["loads_local"] for p in parts), full["loads_local"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("partition", ["reference", "cost", "balanced"])
def test_run_solver_shards_cover_the_problem_with_re_based_maps(partition):
    """run_solver's per-rank slice (run_solver._local_shard) for every partition: the ranks' shards hold every stored entry
    exactly once, and each local column sits under the operator of its global column."""
    from dualip_amd.objectives.matching import MatchingInputArgs
    from dualip_amd.projections import create_projection_map
    from dualip_amd.run_solver import _local_shard

    m, n, cut = 7, 40, 24
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 4, n)
    cp = np.zeros(n + 1, dtype=np.int64)
    cp[1:] = np.cumsum(lens)
    rows = np.concatenate([np.sort(rng.choice(m, k, replace=False)) for k in lens]).astype(np.int64)
    vals = torch.arange(len(rows), dtype=torch.float64)  # value = position: tells which global entry a local one is
    A = torch.sparse_csc_tensor(torch.from_numpy(cp), torch.from_numpy(rows), vals, size=(m, n))
    pm = {**create_projection_map("box", {"lower": 0.0, "upper": 1.0}, None, indices=range(0, cut)), **create_projection_map("simplex", {"z": 1.0}, None, indices=range(cut, n))}
    args = MatchingInputArgs(A=A, c=A, projection_map=pm, b_vec=torch.zeros(m, dtype=torch.float64))
    for world in (1, 3, 4):
        seen = []
        for r in range(world):
            loc = _local_shard(args, r, world, "cpu", partition)
            seen.append(loc.A.values())
            assert sorted(i for e in loc.projection_map.values() for i in e.indices) == list(range(loc.A.size(1)))
            ptr = loc.A.ccol_indices()
            for key, e in loc.projection_map.items():
                for j in e.indices:
                    for v in loc.A.values()[int(ptr[j]) : int(ptr[j + 1])].tolist():
                        col = int(np.searchsorted(cp, int(v), side="right") - 1)
                        assert (col >= cut) == key.startswith("simplex"), (partition, key, col)
            if partition != "balanced":
                v = loc.A.values()
                assert v.numel() == 0 or torch.equal(v, torch.arange(int(v[0]), int(v[0]) + v.numel(), dtype=v.dtype))  # contiguous
        assert torch.equal(torch.cat(seen).sort().values, vals)
