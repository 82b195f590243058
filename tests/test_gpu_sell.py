"""Column-per-lane slices (csrc/sell.h): the short columns of simplex entries, re-laid at handle creation so that one lane
owns one column.  Parity of everything that can meet the slices: the reference goldens, the window-tile path (the same
handle built with DUALIP_HIP_SELL=0), the primal written back in the caller's order, ragged lengths up to and past the
tallest slice, the hot-rows plan, the fairness stream, simplex_eq with and without the reference's padded blocks, cost updates."""
import os

import numpy as np
import pytest
import torch

import oracle
from tests.helpers import NP_DT, RTOL, load, problem, relerr, torch_args

pytestmark = [pytest.mark.gpu]
DEV = "cuda:0"
TD = {"f32": torch.float32, "f64": torch.float64}


def _objective(p, dn, pm, gamma, sell=True, ctor=None, **env):
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction

    env.setdefault("DUALIP_HIP_SELL_MIN_SHARE", 0)  # (an entry is only sliced when most of its non-zeros sit in short columns: not these test shapes)
    old = {k: os.environ.get(k) for k in ["DUALIP_HIP_SELL", *env]}
    os.environ["DUALIP_HIP_SELL"] = "1" if sell else "0"
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        return MatchingSolverDualObjectiveFunction(torch_args(p, dn, pm, DEV), gamma, **(ctor or {}))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _ragged(seed, n=6000, m=300, lens=None):
    rng = np.random.default_rng(seed)
    if lens is None:
        lens = rng.integers(0, 41, n)  # empty columns, every slice height, and columns past the tallest slice (24)
        lens[::50] = 0
    lens = np.minimum(lens, m).astype(np.int64)
    colptr = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=colptr[1:])
    rowidx = np.concatenate([np.sort(rng.choice(m, size=int(k), replace=False)) for k in lens]).astype(np.int64)
    c = -np.minimum(rng.lognormal(-2.0, 0.75, rowidx.shape[0]), 0.5)
    a = -c * rng.lognormal(0.0, 0.5, rowidx.shape[0])
    return dict(m=m, n=len(lens), colptr=colptr, rowidx=rowidx, a=a, c=c, b=rng.uniform(0.5, 3.0, m))


def _boundary_lengths(seed, n=3000, m=600):
    """Column lengths that hit every class of the K-lanes-per-column slices and its edges (24|25, 32|33, 64|65, 128|129, 255|256, 512|513),
    the window limit (253) and columns past everything, plus a random filling of 0 .. 400."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 401, n)
    edges = [1, 2, 23, 24, 25, 26, 31, 32, 33, 34, 47, 48, 49, 63, 64, 65, 66, 96, 127, 128, 129, 130, 192, 252, 253, 254, 255, 256, 257, 287, 288, 289, 300, 399, 448, 511, 512, 513, 577, 600]
    lens[: 3 * len(edges)] = np.repeat(edges, 3)
    rng.shuffle(lens)
    lens[::97] = 0
    return lens


@pytest.mark.parametrize("merge", [0, 1])
@pytest.mark.parametrize("dn", ["f32", "f64"])
@pytest.mark.parametrize("kind", ["simplex", "simplex_eq", "mixed"])
def test_lane_slices_every_length_class(kind, dn, merge):
    """Columns of 25 .. 512 non-zeros are dealt to K = 2 .. 32 lanes each (csrc/sell.h): x, gradient and objective against the
    oracle and against the same handle without slices, at every class edge; the primal comes back in the caller's order.
    merge = 1 (the default for a handle this small): the few short columns join the two-lane class; merge = 0: they keep their one-lane
    slices, so both slice loops of the second binary run in one launch."""
    from dualip_amd.projections import create_projection_map

    p = _ragged(23, m=600, lens=_boundary_lengths(29))
    n, m = p["n"], p["m"]
    lens = np.diff(p["colptr"])
    if kind == "mixed":
        cut = n // 3
        pm = {**create_projection_map("box", {"lower": 0.0, "upper": 1.0}, n, indices=range(cut)), **create_projection_map("simplex", {"z": 2.0}, n, indices=range(cut, n))}
        projs, col_proj, in_entry = [("box", {"lower": 0.0, "upper": 1.0}), ("simplex", {"z": 2.0})], np.r_[np.zeros(cut, np.int32), np.ones(n - cut, np.int32)], lens[cut:]
    else:
        pm = create_projection_map(kind, {"z": 2.0}, n)
        projs, col_proj, in_entry = [(kind, {"z": 2.0})], None, lens
    gamma = 0.05
    ctor = dict(batching=False, simplex_eq_padding="reference") if kind == "simplex_eq" else None
    f = _objective(p, dn, pm, gamma, ctor=ctor, DUALIP_HIP_SELL_MERGE_SHORT=merge)
    f0 = _objective(p, dn, pm, gamma, sell=False, ctor=ctor)
    f1 = _objective(p, dn, pm, gamma, ctor=ctor, DUALIP_HIP_SELL_LANES=0)  # one lane per column only: the longer columns walk alone
    info, info1 = f.info(), f1.info()
    assert info["slice_columns"] == int(((in_entry >= 1) & (in_entry <= 512)).sum())
    assert info["slice_lane_columns"] == int(((in_entry >= (1 if merge else 25)) & (in_entry <= 512)).sum())
    assert info["long_columns"] >= int((in_entry > 512).sum())
    assert info1["slice_lane_columns"] == 0 and info1["slice_columns"] == int(((in_entry >= 1) & (in_entry <= 24)).sum())
    rng = np.random.default_rng(7)
    for scale in (0.0, 0.02, 0.5):
        lam = torch.from_numpy(rng.uniform(0, scale, m) if scale else np.zeros(m)).to(TD[dn]).to(DEV)
        r, r0, r1 = (g.calculate(lam, gamma, save_primal=True) for g in (f, f0, f1))
        ax, obj0, ssq, xo = oracle.matching_calculate(m, n, p["colptr"], p["rowidx"], p["a"], p["c"], lam.cpu().numpy(), gamma, projs, col_proj=col_proj, dtype=NP_DT[dn])
        x = r.primal_var.cpu().numpy()
        assert relerr(x, xo) < RTOL[dn]
        assert relerr(r.dual_gradient.cpu().numpy(), ax - p["b"]) < RTOL[dn]
        for other in (r0, r1):
            assert relerr(x, other.primal_var.cpu().numpy()) < RTOL[dn]
            assert relerr(r.dual_gradient.cpu().numpy(), other.dual_gradient.cpu().numpy()) < RTOL[dn]
            assert abs(float(r.dual_objective) - float(other.dual_objective)) <= RTOL[dn] * 10 * max(1.0, abs(float(other.dual_objective)))
        assert torch.equal(f.calculate(lam, gamma, save_primal=True).primal_var, r.primal_var)  # bit-reproducible
        if kind == "simplex":  # every projected column sums to z or less, and to z where the clamped column exceeded it
            sums = np.add.reduceat(x, p["colptr"][:-1][lens > 0])
            assert (sums <= 2.0 * (1 + (5e-5 if dn == "f32" else 1e-10))).all()  # (fp32: up to 600 terms per column)
    if kind == "simplex_eq":  # exact mode
        fe = _objective(p, dn, pm, gamma)
        lam = torch.from_numpy(rng.uniform(0, 0.3, m)).to(TD[dn]).to(DEV)
        xe = fe.calculate(lam, gamma, save_primal=True).primal_var.cpu().numpy()
        sums = np.add.reduceat(xe, p["colptr"][:-1][lens > 0])
        assert np.allclose(sums, 2.0, atol=2e-4 if dn == "f32" else 1e-9)


@pytest.mark.parametrize("dn", ["f32", "f64"])
@pytest.mark.parametrize("kind", ["simplex", "simplex_eq", "mixed"])
def test_slices_against_oracle_and_window_tiles(kind, dn):
    from dualip_amd.projections import create_projection_map

    p = _ragged(3)
    n, m = p["n"], p["m"]
    if kind == "mixed":
        cut = n // 3
        pm = {**create_projection_map("box", {"lower": 0.0, "upper": 1.0}, n, indices=range(cut)), **create_projection_map("simplex", {"z": 1.5}, n, indices=range(cut, n))}
        projs, col_proj = [("box", {"lower": 0.0, "upper": 1.0}), ("simplex", {"z": 1.5})], np.r_[np.zeros(cut, np.int32), np.ones(n - cut, np.int32)]
    else:
        pm = create_projection_map(kind, {"z": 1.5}, n)
        projs, col_proj = [(kind, {"z": 1.5})], None
    gamma = 0.05
    # simplex_eq: the oracle restates the reference's zero-padded blocks (one block per entry with batching=False); the exact
    # projection (the default) is compared with the window tiles and through its defining property below
    ctor = dict(batching=False, simplex_eq_padding="reference") if kind == "simplex_eq" else None
    f = _objective(p, dn, pm, gamma, ctor=ctor)
    f0 = _objective(p, dn, pm, gamma, sell=False, ctor=ctor)
    info = f.info()
    lens = np.diff(p["colptr"])
    in_entry = lens[n // 3 :] if kind == "mixed" else lens
    assert info["slices"] > 0 and info["slice_columns"] == int((in_entry >= 1).sum())  # (lengths up to 40: one lane per column to 24, two / four beyond)
    assert info["slice_lane_columns"] == int((in_entry >= 1).sum())  # (a handle this small: the short columns join the two-lane class)
    assert f0.info()["slices"] == 0
    rng = np.random.default_rng(5)
    for scale in (0.0, 0.02, 0.5):
        lam = torch.from_numpy(rng.uniform(0, scale, m) if scale else np.zeros(m)).to(TD[dn]).to(DEV)
        r = f.calculate(lam, gamma, save_primal=True)
        x = r.primal_var.clone()
        r0 = f0.calculate(lam, gamma, save_primal=True)
        ax, obj0, ssq, xo = oracle.matching_calculate(m, n, p["colptr"], p["rowidx"], p["a"], p["c"], lam.cpu().numpy(), gamma, projs, col_proj=col_proj, dtype=NP_DT[dn])
        assert relerr(x.cpu().numpy(), xo) < RTOL[dn]
        assert relerr(r.dual_gradient.cpu().numpy(), ax - p["b"]) < RTOL[dn]
        assert relerr(x.cpu().numpy(), r0.primal_var.cpu().numpy()) < RTOL[dn]
        assert relerr(r.dual_gradient.cpu().numpy(), r0.dual_gradient.cpu().numpy()) < RTOL[dn]
        assert abs(float(r.dual_objective) - float(r0.dual_objective)) <= RTOL[dn] * 10 * max(1.0, abs(float(r0.dual_objective)))
        # bit-reproducible: a second launch, and a second handle
        assert torch.equal(f.calculate(lam, gamma, save_primal=True).primal_var, x)
    if kind == "simplex_eq":  # exact mode: every non-empty column sums to z, and the two paths agree
        fe, fe0 = _objective(p, dn, pm, gamma), _objective(p, dn, pm, gamma, sell=False)
        lam = torch.from_numpy(rng.uniform(0, 0.3, m)).to(TD[dn]).to(DEV)
        xe = fe.calculate(lam, gamma, save_primal=True).primal_var.cpu().numpy()
        assert relerr(xe, fe0.calculate(lam, gamma, save_primal=True).primal_var.cpu().numpy()) < RTOL[dn]
        sums = np.add.reduceat(xe, p["colptr"][:-1][lens > 0])
        assert np.allclose(sums, 1.5, atol=1e-4 if dn == "f32" else 1e-9)
    f2 = _objective(p, dn, pm, gamma, ctor=ctor)
    lam = torch.from_numpy(rng.uniform(0, 0.1, m)).to(TD[dn]).to(DEV)
    ra, rb = f.calculate(lam, gamma, save_primal=True), f2.calculate(lam, gamma, save_primal=True)
    assert torch.equal(ra.dual_gradient, rb.dual_gradient) and torch.equal(ra.primal_var, rb.primal_var) and float(ra.dual_objective) == float(rb.dual_objective)


def test_slices_on_reference_goldens():
    """The g1 goldens (reference's own calculate) with the slices on: every simplex column of that problem is short."""
    from dualip_amd.projections import create_projection_map

    z = load("g1_syn2000.npz")
    p = problem(z)
    checked = 0
    for dn in ("f32", "f64"):
        for mn, (ptype, params) in {"simplex1": ("simplex", {"z": 1.0}), "simplex2.5": ("simplex", {"z": 2.5})}.items():
            gammas = sorted({k.split("|")[1] for k in z.files if k.startswith(mn + "|")})
            for gs in gammas:
                gamma = float(gs)
                f = _objective(p, dn, create_projection_map(ptype, params, p["n"]), gamma)
                assert f.info()["slices"] > 0 and f.info()["tiles"] == f.info()["long_columns"]
                for ln in ("zero", "small", "large"):
                    key = f"{mn}|{gs}|{ln}|{dn}"
                    if key + "|x" not in z.files:
                        continue
                    r = f.calculate(torch.from_numpy(z[f"lam_{ln}"]).to(TD[dn]).to(DEV), gamma, save_primal=True)
                    assert relerr(r.primal_var.cpu().numpy(), z[key + "|x"]) < RTOL[dn], key
                    assert relerr(r.dual_gradient.cpu().numpy(), z[key + "|grad"]) < RTOL[dn], key
                    checked += 1
    assert checked >= 8


def test_slices_with_hot_rows_plan_and_solver_loop():
    """More rows than the LDS holds (hot-rows plan) + slices, through the device-resident loop, against the oracle's loop."""
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent
    from dualip_amd.projections import create_projection_map
    from oracle import agd_oracle

    p = _ragged(11, n=20000, m=2000)
    n, m = p["n"], p["m"]
    pm = create_projection_map("simplex", {"z": 1.0}, n)
    gamma, iters = 0.05, 40
    f = _objective(p, "f64", pm, gamma, DUALIP_HIP_HOT_ROWS=1024)
    assert f.info()["hot_rows"] == 1024 and f.info()["slices"] > 0
    res = AcceleratedGradientDescent(max_iter=iters, gamma=gamma, initial_step_size=1e-3, max_step_size=0.1, save_primal=True, iteration_callback=False).maximize(
        f, torch.zeros(m, dtype=torch.float64, device=DEV))

    def calc(lam, g):
        ax, obj0, ssq, x = oracle.matching_calculate(m, n, p["colptr"], p["rowidx"], p["a"], p["c"], lam, g, [("simplex", {"z": 1.0})])
        grad, obj, *_ = agd_oracle.epilogue(ax, obj0, ssq, lam, p["b"], g, np.float64)
        return grad, obj, x

    want = agd_oracle.maximize(calc, np.zeros(m), iters, gamma, 1e-3, 0.1)
    assert relerr(np.array(res.dual_objective_log)[:30], want["dual_obj_log"][:30]) < 1e-9
    assert relerr(res.objective_result.primal_var.cpu().numpy(), want["last"][2]) < 1e-6


def test_cost_update_refreshes_the_slices():
    from dualip_amd.projections import create_projection_map

    p = _ragged(17, n=4000, m=200)
    pm = create_projection_map("simplex", {"z": 1.0}, p["n"])
    f = _objective(p, "f64", pm, 0.05)
    lam = torch.full((p["m"],), 0.01, dtype=torch.float64, device=DEV)
    f.c.values().mul_(1.7)
    f.costs_changed()
    p2 = dict(p, c=p["c"] * 1.7)
    _, _, _, xo = oracle.matching_calculate(p["m"], p["n"], p["colptr"], p["rowidx"], p["a"], p2["c"], lam.cpu().numpy(), 0.05, [("simplex", {"z": 1.0})])
    assert relerr(f.calculate(lam, 0.05, save_primal=True).primal_var.cpu().numpy(), xo) < 1e-9


def test_value_update_refreshes_the_slices():
    """A and c rewritten in place (same pattern): ``values_changed()`` refreshes the handle's transposed copies and scales, so
    sliced columns and window tiles see the same data again."""
    from dualip_amd.projections import create_projection_map

    p = _ragged(19, n=4000, m=200)
    half = p["n"] // 2
    pm = {**create_projection_map("box", {"lower": 0.0, "upper": 1.0}, p["n"], indices=range(half)), **create_projection_map("simplex", {"z": 1.0}, p["n"], indices=range(half, p["n"]))}
    f = _objective(p, "f64", pm, 0.05)
    lam = torch.full((p["m"],), 0.01, dtype=torch.float64, device=DEV)
    f.A.values().mul_(2.5)
    f.c.values().mul_(0.6)
    f.values_changed()
    col_proj = np.zeros(p["n"], dtype=np.int32)
    col_proj[half:] = 1
    ax, _, _, xo = oracle.matching_calculate(p["m"], p["n"], p["colptr"], p["rowidx"], p["a"] * 2.5, p["c"] * 0.6, lam.cpu().numpy(), 0.05,
                                            [("box", {"lower": 0.0, "upper": 1.0}), ("simplex", {"z": 1.0})], col_proj=col_proj)
    res = f.calculate(lam, 0.05, save_primal=True)
    assert relerr(res.primal_var.cpu().numpy(), xo) < 1e-9
    assert relerr(res.dual_gradient.cpu().numpy() + p["b"], ax) < 1e-9


def test_bisection_entries_take_the_dense_block_route():
    """method="bisection_search" inside a matching objective: not substituted by the exact projection -- its columns go through
    the operator itself (dense blocks), so x equals what the reference's operator returns for them."""
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.projections import create_projection_map, project

    p = _ragged(23, n=800, m=60, lens=np.random.default_rng(1).integers(1, 9, 800))
    n, m = p["n"], p["m"]
    pm = create_projection_map("simplex", {"z": 1.0, "method": "bisection_search"}, n)
    f = MatchingSolverDualObjectiveFunction(torch_args(p, "f64", pm, DEV), 0.05, batching=False)
    assert f._custom is not None and f.info()["slices"] == 0
    lam = torch.full((m,), 0.02, dtype=torch.float64, device=DEV)
    x = f.calculate(lam, 0.05, save_primal=True).primal_var.cpu().numpy()
    # the same columns as one zero-padded block through the operator
    lens = np.diff(p["colptr"])
    L = int(lens.max())
    v = p["a"] * (-1.0 / 0.05 * 0.02) + (-1.0 / 0.05) * p["c"]
    block = np.zeros((L, n))
    cols = np.repeat(np.arange(n), lens)
    offs = np.arange(len(v)) - np.repeat(p["colptr"][:-1], lens)
    block[offs, cols] = v
    want = project("simplex", z=1.0, method="bisection_search")(torch.from_numpy(block).to(DEV)).cpu().numpy()[offs, cols]
    assert relerr(x, want) < 1e-12
    exact = project("simplex", z=1.0)(torch.from_numpy(block).to(DEV)).cpu().numpy()[offs, cols]
    assert np.abs(want - exact).max() > 1e-8  # the two methods do differ (bracket width) -- so the route matters


@pytest.mark.parametrize("kind", ["box", "mixed", "simplex"])
def test_device_packed_windows_equal_host_packed(kind):
    """Window tiles packed on the device (chunks of 8192 columns, one thread each) against the host's greedy packing of the same
    columns: the exact integer gradient is bit-identical, the primal identical, the tile count within a chunk-boundary's worth."""
    from dualip_amd.projections import create_projection_map

    rng = np.random.default_rng(9)
    n, m = 40_000, 500
    lens = rng.poisson(9, n)
    lens[rng.integers(0, n, 40)] = rng.integers(260, 400, 40)  # single-column tiles
    lens[::333] = 0
    p = _ragged(9, n=n, m=m, lens=lens)
    if kind == "mixed":
        pm = {**create_projection_map("box", {"lower": 0.0, "upper": 1.0}, n, indices=range(n // 2)), **create_projection_map("simplex", {"z": 1.0}, n, indices=range(n // 2, n))}
    else:
        pm = create_projection_map(kind, {"z": 1.0} if kind == "simplex" else {"lower": 0.0, "upper": 1.0}, n)
    fd = _objective(p, "f64", pm, 0.05, DUALIP_HIP_SELL_MIN_SHARE=0.5)
    fh = _objective(p, "f64", pm, 0.05, DUALIP_HIP_SELL_MIN_SHARE=0.5, DUALIP_HIP_HOST_PACK=1)
    a, b = fd.info(), fh.info()
    assert a["slices"] == b["slices"] and a["long_columns"] == b["long_columns"]
    assert b["tiles"] <= a["tiles"] <= b["tiles"] + n // 8192 + 2
    lam = torch.from_numpy(rng.uniform(0, 0.2, m)).to(DEV)
    ra, rb = fd.calculate(lam, 0.05, save_primal=True), fh.calculate(lam, 0.05, save_primal=True)
    assert torch.equal(ra.dual_gradient, rb.dual_gradient) and torch.equal(ra.primal_var, rb.primal_var)
    assert abs(float(ra.dual_objective) - float(rb.dual_objective)) < 1e-9 * max(1.0, abs(float(rb.dual_objective)))


def test_fairness_stream_on_sliced_handle_equals_window_handle():
    """dl_matching_set_fairness on a handle WITH slices (the C path keeps a transposed copy of f) against the same handle kept on
    window tiles (what objectives/matching_fairness.py builds): same gradient (exact integers), same primal."""
    if os.environ.get("DUALIP_HIP_SELL") == "0":
        pytest.skip("needs a handle with slices")
    from dualip_amd import _hip
    from dualip_amd.objectives.matching import MatchingInputArgs, MatchingSolverDualObjectiveFunction
    from dualip_amd.projections import create_projection_map

    p = _ragged(31, n=5000, m=120, lens=np.random.default_rng(2).integers(1, 20, 5000))
    n, K = p["n"], p["m"]
    m = K + 2
    rng = np.random.default_rng(4)
    f_vals = torch.from_numpy(rng.uniform(-0.3, 0.3, len(p["a"]))).to(DEV)
    out = []
    for slices in (True, False):
        base = torch_args(p, "f64", create_projection_map("simplex", {"z": 1.0}, n), DEV)
        wide = lambda t: torch.sparse_csc_tensor(t.ccol_indices(), t.row_indices(), t.values(), size=(m, n))  # noqa: E731
        b = torch.cat([base.b_vec, torch.tensor([0.05, 0.05], dtype=torch.float64, device=DEV)])
        obj = MatchingSolverDualObjectiveFunction(MatchingInputArgs(A=wide(base.A), c=wide(base.c), projection_map=base.projection_map, b_vec=b), 0.05, column_slices=slices)
        assert (obj.info()["slices"] > 0) == slices
        with torch.cuda.device(obj.device):
            _hip.check(_hip.load().dl_matching_set_fairness(obj._handle, _hip.ptr(f_vals), _hip.stream_ptr(obj.device)))
        lam = torch.from_numpy(rng.uniform(0, 0.1, m)).to(DEV) if not out else out[0][2]
        r = obj.calculate(lam, 0.05, save_primal=True)
        out.append((r.dual_gradient.clone(), r.primal_var.clone(), lam))
    assert relerr(out[0][1].cpu().numpy(), out[1][1].cpu().numpy()) < 1e-12
    assert relerr(out[0][0].cpu().numpy(), out[1][0].cpu().numpy()) < 1e-12
