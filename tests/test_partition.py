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


def test_generic_lp_is_sharded_by_variables_and_box_defaults_follow_the_reference():
    """Host logic of BASELINE config 5 on several ranks (run_solver._local_lp_shard): the ranks' shards are a partition of the VARIABLES in the
    reference's contiguous sizes n // W (+1 for the first n % W), every layout of A gives the same columns, c / projection entries are re-based,
    b_vec and the equality mask stay whole.  And the bound semantics the kernel's clamp arrays are built from (objectives/miplib._box_bounds): a
    box entry that does not name a bound keeps BoxProjection's default for it (box.py:12-13: ``{"upper": 1}`` is [0, 1]); ``l`` / ``u`` entries
    (the reference's bound reader, miplib.py:111-121) treat a missing key as "no bound"; NaN / None is an absent bound."""
    from dualip_amd.objectives.miplib import MIPLIBInputArgs, _box_bounds
    from dualip_amd.projections.base import ProjectionEntry
    from dualip_amd.run_solver import _local_lp_shard

    inf = float("inf")
    assert _box_bounds({}) == (0.0, 1.0) and _box_bounds({"upper": 1}) == (0.0, 1.0) and _box_bounds({"lower": -2}) == (-2.0, 1.0)
    assert _box_bounds({"l": 0.0, "u": float("nan")}) == (0.0, inf) and _box_bounds({"l": 0.0}) == (0.0, inf) and _box_bounds({"u": 5}) == (-inf, 5.0)
    assert _box_bounds({"lower": float("nan"), "upper": 3}) == (-inf, 3.0) and _box_bounds({"lower": 0.25, "upper": 2.0}) == (0.25, 2.0)

    g = torch.Generator().manual_seed(3)
    m, n = 7, 23
    A = torch.where(torch.rand(m, n, generator=g) < 0.4, torch.randn(m, n, generator=g), torch.zeros(m, n)).double()
    c, b = torch.randn(n, generator=g).double(), torch.rand(m, generator=g).double()
    eq = torch.zeros(m, dtype=torch.bool)
    eq[2] = True
    pm = {"a": ProjectionEntry("box", {"lower": 0.0, "upper": 2.0}, indices=list(range(0, 9))),
          "b": ProjectionEntry("cone", {"lower": 0.0}, indices=[9, 11, 13, 20, 22]),
          "c": ProjectionEntry("box", {}, indices=range(14, 20))}
    for world in (1, 2, 3, 8, 30):  # (30 ranks for 23 variables: seven empty shards)
        for form in ("dense", "coo", "csr", "csc"):
            Af = {"dense": A, "coo": A.to_sparse_coo(), "csr": A.to_sparse_csr(), "csc": A.to_sparse_csc()}[form]
            args = MIPLIBInputArgs(A=Af, c=c, projection_map=pm, b_vec=b, equality_mask=eq)
            cols, seen = [], {k: [] for k in pm}
            for rank in range(world):
                loc = _local_lp_shard(args, rank, world, "cpu")
                width = loc.A.shape[1]
                want = n // world + (1 if rank < n % world else 0)
                assert width == want and loc.c.shape == (width,) and torch.equal(loc.b_vec, b) and torch.equal(loc.equality_mask, eq)
                lo = sum(n // world + (1 if r < n % world else 0) for r in range(rank))
                dense = loc.A.to_dense() if loc.A.layout != torch.strided else loc.A
                assert torch.equal(dense, A[:, lo : lo + width]) and torch.equal(loc.c, c[lo : lo + width])
                cols += list(range(lo, lo + width))
                for k, e in loc.projection_map.items():
                    assert e.proj_type == pm[k].proj_type and e.proj_params == pm[k].proj_params
                    seen[k] += [int(i) + lo for i in e.indices]
            assert cols == list(range(n))
            for k in pm:
                assert sorted(seen[k]) == sorted(int(i) for i in pm[k].indices), (world, form, k)
