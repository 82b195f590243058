"""Synthetic matching LP generator, on the device (the data side of BASELINE configs 2-4).

Follows the generative model of the reference's benchmark/generate_synthetic_data.py:27-164 -- destination breadth
Z_j ~ LogN(0,1) with p_j = Z_j / sum Z * (sparsity * m), scale s_j ~ LogN(0,1), base value v_j ~ LogN(-4, .75), source
affinity u_i ~ LogN(0,.5), edge noise eps ~ LogN(0,.5); c_ij = min(v_j u_i eps, 0.5) stored negated; a_ij = s_j c_ij;
b_j = rho_j (greedy_load_j + 1e-8), rho ~ U(.5,1), greedy load = each source assigned to its largest-a edge --
but draws the edges per SOURCE so that it vectorises and shards by columns: deg_i ~ Poisson(sum_j p_j), destinations
i.i.d. ~ Categorical(p) with duplicates removed.  (The reference draws K_j ~ Poisson(p_j n) sources per destination
in a Python loop: 8.6 s per 1M sources, >50 GB at 100M.  Both give edge (i, j) with probability ~ p_j, independent
across pairs; the two generators are equal in distribution to first order in p_j, not bit-identical.)

Columns are produced in fixed-size chunks with per-chunk seeds, so rank r of W can build exactly its own column
range of the SAME global problem; only the m-sized greedy loads need one all-reduce.
"""
import numpy as np
import torch

CHUNK_COLS = 250_000


def _dest_params(num_destinations, target_sparsity, seed):
    rng = np.random.default_rng(seed)
    Z = rng.lognormal(0.0, 1.0, num_destinations)
    p = Z / Z.sum() * (target_sparsity * num_destinations)
    s = rng.lognormal(0.0, 1.0, num_destinations)
    v = rng.lognormal(-4.0, 0.75, num_destinations)
    rho = rng.uniform(0.5, 1.0, num_destinations)
    return p, s, v, rho


def _chunk(c0, c1, m, p_cdf, p_total, s_t, v_t, seed, device, keep=None):
    """Edges of the sources [c0, c1) of one generator chunk (c0 a multiple of CHUNK_COLS, c1 the chunk's or the problem's end),
    restricted to the global columns keep = (lo, hi) when given: returns (counts[int64], rows[int32], a[f32], c[f32], loads)."""
    g = torch.Generator(device=device)
    g.manual_seed((int(seed) * 1_000_003 + c0 // CHUNK_COLS) % (2**63 - 1))
    ncol = c1 - c0
    deg = torch.poisson(torch.full((ncol,), p_total, device=device, dtype=torch.float32), generator=g).to(torch.int64)
    total = int(deg.sum())
    col = torch.repeat_interleave(torch.arange(ncol, device=device, dtype=torch.int64), deg, output_size=total)
    dest = torch.searchsorted(p_cdf, torch.rand(total, device=device, dtype=torch.float64, generator=g)).clamp_(max=m - 1)
    key = torch.unique(col * m + dest, sorted=True)  # sorted by (source, destination), duplicates dropped
    col = torch.div(key, m, rounding_mode="floor")
    dest = key - col * m
    u = torch.exp(torch.randn(ncol, device=device, generator=g) * 0.5)
    eps = torch.exp(torch.randn(key.numel(), device=device, generator=g) * 0.5)
    c = torch.clamp(v_t[dest] * u[col] * eps, max=0.5)
    a = s_t[dest] * c
    counts = torch.bincount(col, minlength=ncol)
    if keep is not None and (keep[0] > c0 or keep[1] < c1):
        # a shard whose cut falls inside this chunk: the WHOLE chunk is drawn (its random stream belongs to the chunk), then only
        # the columns [keep[0], keep[1]) are kept -- edges are sorted by source, so that is one contiguous run of them
        k_lo, k_hi = max(keep[0], c0) - c0, min(keep[1], c1) - c0
        ends = torch.cumsum(counts, 0)
        e0 = int(ends[k_lo - 1]) if k_lo > 0 else 0
        e1 = int(ends[k_hi - 1]) if k_hi > 0 else 0
        col, dest, a, c, counts = col[e0:e1] - k_lo, dest[e0:e1], a[e0:e1], c[e0:e1], counts[k_lo:k_hi]
        ncol = k_hi - k_lo
    # greedy load: every source contributes its largest a to that edge's destination
    best = torch.zeros(ncol, device=device, dtype=a.dtype).scatter_reduce_(0, col, a, reduce="amax", include_self=False)
    is_best = a == best[col]
    loads = torch.zeros(m, device=device, dtype=torch.float64).index_add_(0, dest[is_best], a[is_best].double())
    return counts, dest.to(torch.int32), a, -c, loads


def generate_matching_problem(num_sources, num_destinations, target_sparsity, seed=42, device="cuda:0", dtype=torch.float32, col_range=None, reduce_loads=None,
                              col_ranges=None):
    """Build the (shard of the) synthetic problem on ``device``.

    col_range=(lo, hi): generate only these global columns (any cut points: a chunk a cut falls into is drawn whole and sliced,
    so every shard of every partition sees the SAME global problem).
    col_ranges=[(lo, hi), ...]: several such ranges, concatenated in the given order (a shard that takes its share of
    every projection block).
    reduce_loads: callable applied to the float64[m] greedy-load vector before b is formed (pass an all-reduce when
    sharded).  Returns dict(input_args=MatchingInputArgs(projection_map=None -- set by the caller), nnz, m, n_local,
    loads_local = this call's greedy loads before reduce_loads, rho).
    """
    from dualip_amd.objectives.matching import MatchingInputArgs

    m = int(num_destinations)
    if col_ranges is None:
        col_ranges = [(0, int(num_sources)) if col_range is None else (int(col_range[0]), int(col_range[1]))]
    col_ranges = [(int(a), int(b)) for a, b in col_ranges if int(b) > int(a)]
    n_local = sum(b - a for a, b in col_ranges)
    p, s, v, rho = _dest_params(m, target_sparsity, seed)
    p_t = torch.from_numpy(p).to(device)
    p_cdf = torch.cumsum(p_t / p_t.sum(), 0)
    s_t = torch.from_numpy(s).to(device=device, dtype=torch.float32)
    v_t = torch.from_numpy(v).to(device=device, dtype=torch.float32)
    counts, rows, a_parts, c_parts = [], [], [], []
    loads = torch.zeros(m, device=device, dtype=torch.float64)
    n_global = int(num_sources)
    for lo, hi in col_ranges:
        c0 = lo // CHUNK_COLS * CHUNK_COLS
        while c0 < hi:
            c1 = min(n_global, c0 + CHUNK_COLS)  # (a chunk is always drawn whole: cuts inside a chunk keep a slice of it)
            cnt, r, a, c, ld = _chunk(c0, c1, m, p_cdf, float(p.sum()), s_t, v_t, seed, device, keep=(lo, hi))
            counts.append(cnt)
            rows.append(r)
            a_parts.append(a)
            c_parts.append(c)
            loads += ld
            c0 = c1
    counts = torch.cat(counts) if counts else torch.zeros(0, dtype=torch.int64, device=device)
    nnz = int(counts.sum())
    idx_dtype = torch.int32 if nnz < 2**31 - 1 else torch.int64
    colptr = torch.zeros(n_local + 1, dtype=idx_dtype, device=device)
    colptr[1:] = torch.cumsum(counts, 0).to(idx_dtype)
    del counts
    rowidx = torch.cat(rows).to(idx_dtype) if rows else torch.zeros(0, dtype=idx_dtype, device=device)
    del rows
    a_vals = torch.cat(a_parts).to(dtype) if a_parts else torch.zeros(0, dtype=dtype, device=device)
    del a_parts
    c_vals = torch.cat(c_parts).to(dtype) if c_parts else torch.zeros(0, dtype=dtype, device=device)
    del c_parts
    loads_local = loads.clone()
    if reduce_loads is not None:
        loads = reduce_loads(loads)
    b = (torch.from_numpy(rho).to(device) * (loads + 1e-8)).to(dtype)
    A = torch.sparse_csc_tensor(colptr, rowidx, a_vals, size=(m, n_local), check_invariants=False)
    C = torch.sparse_csc_tensor(colptr, rowidx, c_vals, size=(m, n_local), check_invariants=False)
    args = MatchingInputArgs(A=A, c=C, projection_map=None, b_vec=b, equality_mask=None)
    return dict(input_args=args, nnz=nnz, m=m, n_local=n_local, loads_local=loads_local, rho=rho)
