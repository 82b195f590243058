"""GPU edge cases of the fused matching pass against the CPU oracle (``-m gpu``): shapes the golden fixtures do not reach.

  * more than 65 536 dual rows  -> 32-bit row indices, the dual vector and the gradient no longer fit the LDS (global-atomic plan)
  * more than 255 projection entries -> entries beyond the LDS table are served by the single-column path
  * value arrays that are not 16-byte aligned, and tiny problems -> aligned, zero-padded copies owned by the handle
  * columns longer than a 256-element window, empty columns, an all-empty problem, a column range of one operator inside another
Tolerance: RTOL of tests/helpers.py (2e-4 fp32 / 1e-9 fp64, relative to the largest magnitude).
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import agd_oracle
from tests.helpers import NP_DT, RTOL, relerr, torch_args

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _random_problem(m, n, mean_deg, seed, long_cols=(), empty_every=0):
    rng = np.random.default_rng(seed)
    deg = rng.poisson(mean_deg, n).astype(np.int64)
    deg = np.minimum(deg, m)
    for j, d in long_cols:
        deg[j] = d
    if empty_every:
        deg[::empty_every] = 0
    colptr = np.zeros(n + 1, dtype=np.int64)
    colptr[1:] = np.cumsum(deg)
    rows = np.concatenate([np.sort(rng.choice(m, size=int(d), replace=False)) for d in deg]) if deg.sum() else np.zeros(0, dtype=np.int64)
    nnz = int(colptr[-1])
    a = rng.uniform(0.05, 1.0, nnz)
    c = -rng.uniform(0.01, 0.5, nnz)
    b = rng.uniform(0.5, 2.0, m)
    return dict(m=m, n=n, colptr=colptr, rowidx=rows.astype(np.int64), a=a, c=c, b=b)


def _compare(p, pm, entries, col_proj, gamma, dn, lam, scale=1.0):
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction

    f = MatchingSolverDualObjectiveFunction(torch_args(p, dn, pm, DEV), gamma=gamma)
    td = torch.float32 if dn == "f32" else torch.float64
    res = f.calculate(torch.from_numpy(lam).to(td).to(DEV), gamma=gamma, save_primal=True)
    ax, obj0, ssq, x = oracle.matching_calculate(p["m"], p["n"], p["colptr"], p["rowidx"], p["a"], p["c"], lam, gamma, entries, col_proj=col_proj, dtype=NP_DT[dn])
    grad, obj, reg, dvtg, mx, sm = agd_oracle.epilogue(ax, obj0, ssq, lam, p["b"], gamma, NP_DT[dn])
    assert relerr(res.dual_gradient.cpu().numpy(), grad) < RTOL[dn] * scale
    assert relerr(res.primal_var.cpu().numpy(), x) < RTOL[dn] * scale
    assert relerr([float(res.dual_objective), float(res.reg_penalty)], [obj, reg]) < RTOL[dn] * 10 * scale
    return f


@pytest.mark.parametrize("hot", [True, False])
def test_more_than_65536_rows(hot, monkeypatch):
    """32-bit row indices; the dual vector and the gradient do not fit the LDS: hot-rows plan, or (plan disabled) global atomics."""
    import os

    from dualip_amd.projections import create_projection_map

    if not hot:
        monkeypatch.setenv("DUALIP_HIP_HOT_ROWS", "0")
    m, n = 70_000, 6_000
    p = _random_problem(m, n, 9, seed=5)
    lam = np.random.default_rng(1).uniform(0, 0.01, m)
    narrow = False  # (one tile layout since round 5)
    for dn in ("f32", "f64"):
        for pt, pp in (("simplex", {"z": 1.0}), ("box", {"lower": 0.0, "upper": 1.0})):
            f = _compare(p, create_projection_map(pt, dict(pp), n), [(pt, pp)], None, 0.05, dn, lam)
            info = f.info()
            assert info["row_index_bytes"] == 4
            if hot and not narrow:
                assert info["hot_rows"] > 0 and info["lambda_in_lds"] == 1 and info["grad_in_lds"] == 1, info
            else:
                assert info["hot_rows"] == 0 and info["lambda_in_lds"] == 0 and info["grad_in_lds"] == 0, info


def test_more_projection_entries_than_the_lds_table():
    from dualip_amd.projections.base import ProjectionEntry

    m, n = 300, 6_000
    p = _random_problem(m, n, 8, seed=9)
    lam = np.random.default_rng(2).uniform(0, 0.02, m)
    pm, entries, col_proj = {}, [], np.full(n, -1, dtype=np.int32)
    per = 15  # 400 entries of 15 columns: box bounds / simplex radii that differ per entry
    for e in range(n // per):
        idx = list(range(e * per, (e + 1) * per))
        if e % 3 == 0:
            kind, params = "simplex", {"z": 0.5 + 0.01 * e}
        elif e % 3 == 1:
            kind, params = "box", {"lower": 0.0, "upper": 0.2 + 0.002 * e}
        else:
            kind, params = "cone", {"lower": 0.001 * e}
        pm[f"e{e}"] = ProjectionEntry(kind, params, indices=idx)
        entries.append((kind, params))
        col_proj[idx] = e
    for dn in ("f32", "f64"):
        _compare(p, pm, entries, col_proj, 0.02, dn, lam)


def test_unaligned_values_and_tiny_problems_are_staged_into_the_same_kernel():
    """Value arrays that are not 16-byte aligned, and problems of fewer than 1024 non-zeros, used to take a second (64-wide) kernel; since
    round 5 the handle reads its own aligned, zero-padded copies and every input takes the 256-wide layout: the unaligned handle returns
    the SAME BITS as the aligned one, value refreshes follow the caller's arrays, tiny problems agree with the oracle."""
    from dualip_amd.objectives.matching import MatchingInputArgs, MatchingSolverDualObjectiveFunction
    from dualip_amd.projections import create_projection_map

    p = _random_problem(120, 900, 7, seed=3, long_cols=((5, 100), (400, 90)), empty_every=37)
    lam = np.random.default_rng(4).uniform(0, 0.05, p["m"])
    pm = create_projection_map("simplex", {"z": 1.0}, p["n"])
    ref = MatchingSolverDualObjectiveFunction(torch_args(p, "f32", pm, DEV), gamma=0.05)
    assert ref.info()["layout"] == 4
    want = ref.calculate(torch.from_numpy(lam).float().to(DEV), save_primal=True)
    want_grad, want_x, want_obj = want.dual_gradient.clone(), want.primal_var.clone(), float(want.dual_objective)
    # the same values one element into a larger buffer: 4-byte aligned, not 16-byte aligned
    nnz = int(p["colptr"][-1])
    a_buf = torch.zeros(nnz + 8, dtype=torch.float32, device=DEV)
    c_buf = torch.zeros(nnz + 8, dtype=torch.float32, device=DEV)
    a_buf[1 : nnz + 1] = torch.from_numpy(p["a"]).float().to(DEV)
    c_buf[1 : nnz + 1] = torch.from_numpy(p["c"]).float().to(DEV)
    colptr, rowidx = torch.from_numpy(p["colptr"]).to(DEV), torch.from_numpy(p["rowidx"]).to(DEV)
    A = torch.sparse_csc_tensor(colptr, rowidx, a_buf[1 : nnz + 1], size=(p["m"], p["n"]), check_invariants=False)
    C = torch.sparse_csc_tensor(colptr, rowidx, c_buf[1 : nnz + 1], size=(p["m"], p["n"]), check_invariants=False)
    if A.values().data_ptr() % 16 != 0:  # (torch may copy the slice into a fresh, aligned allocation: then there is nothing to test)
        f = MatchingSolverDualObjectiveFunction(MatchingInputArgs(A=A, c=C, projection_map=pm, b_vec=torch.from_numpy(p["b"]).float().to(DEV), equality_mask=None), gamma=0.05)
        assert f.info()["layout"] == 4
        got = f.calculate(torch.from_numpy(lam).float().to(DEV), save_primal=True)
        assert torch.equal(got.dual_gradient, want_grad) and torch.equal(got.primal_var, want_x) and float(got.dual_objective) == want_obj
        # the handle reads COPIES: an in-place change of the caller's arrays reaches it through values_changed(), as for every handle
        a_buf[1 : nnz + 1].mul_(0.5)
        f.values_changed()
        half = f.calculate(torch.from_numpy(lam).float().to(DEV), save_primal=True)
        q = dict(p, a=p["a"] * 0.5)
        ax, obj0, ssq, x = oracle.matching_calculate(q["m"], q["n"], q["colptr"], q["rowidx"], q["a"].astype(np.float32), q["c"], lam, 0.05, [("simplex", {"z": 1.0})], dtype=np.float32)
        grad, obj, *_ = agd_oracle.epilogue(ax, obj0, ssq, lam, q["b"], 0.05, np.float32)
        assert relerr(half.dual_gradient.cpu().numpy(), grad) < RTOL["f32"] and relerr(half.primal_var.cpu().numpy(), x) < RTOL["f32"]
    # fewer than 1024 non-zeros (down to a 5 x 5 problem and an empty one elsewhere in the suite): the same layout on a padded copy
    q = _random_problem(40, 60, 6, seed=8, empty_every=7)
    # release_inputs() on a STAGED handle (nnz < 1024, and nnz % 4 != 0 below): the handle already reads only its own padded copies, so the
    # release is bookkeeping -- same bits before and after, the caller's tensors may go, value refreshes are refused afterwards
    for qq in (q, _random_problem(60, 150, 5, seed=12)):
        assert int(qq["colptr"][-1]) < 1024  # (staged: fewer non-zeros than one round of quads)
        for pt, pp in (("simplex", {"z": 1.0}), ("box", {"lower": 0.0, "upper": 1.0})):
            args = torch_args(qq, "f32", create_projection_map(pt, dict(pp), qq["n"]), DEV)
            f = MatchingSolverDualObjectiveFunction(args, gamma=0.05)
            lam_q = torch.from_numpy(np.random.default_rng(6).uniform(0, 0.05, qq["m"])).float().to(DEV)
            before = f.calculate(lam_q, save_primal=True)
            bg, bx, bo, owned = before.dual_gradient.clone(), before.primal_var.clone(), float(before.dual_objective), f.info()["owned_bytes"]
            out = f.release_inputs()
            del args
            assert out["owned_bytes"] == owned and out["kept_elements"] >= int(qq["colptr"][-1]) and out["kept_elements"] % 4 == 0, out
            junk = torch.full((1 << 16,), 7.0, device=DEV)  # (whatever takes the freed tensors' place)
            after = f.calculate(lam_q, save_primal=True)
            assert torch.equal(after.dual_gradient, bg) and torch.equal(after.primal_var, bx) and float(after.dual_objective) == bo
            with pytest.raises((RuntimeError, ValueError), match="owns its inputs"):
                f.values_changed()
            del junk
    for dn in ("f32", "f64"):
        for pt, pp in (("simplex", {"z": 1.0}), ("box", {"lower": 0.0, "upper": 1.0}), ("simplex_eq", {"z": 1.0})):
            f = _compare(q, create_projection_map(pt, dict(pp), q["n"]), [(pt, pp)], None, 0.05, dn, np.random.default_rng(6).uniform(0, 0.05, q["m"]))
            assert f.info()["layout"] == 4


def test_long_columns_empty_columns_and_nested_ranges():
    from dualip_amd.projections.base import ProjectionEntry

    m, n = 2_000, 4_000
    # columns longer than one 256-element window, next to ordinary and empty ones
    p = _random_problem(m, n, 10, seed=21, long_cols=((1, 300), (17, 1500), (1999, 257), (3999, 700)), empty_every=11)
    lam = np.random.default_rng(7).uniform(0, 0.01, m)
    pm = {
        "simplex_a": ProjectionEntry("simplex", {"z": 1.0}, indices=list(range(0, 1000))),
        "box": ProjectionEntry("box", {"lower": 0.0, "upper": 0.7}, indices=list(range(1000, 1800))),
        "simplex_b": ProjectionEntry("simplex_eq", {"z": 2.0}, indices=list(range(2500, 4000))),
        # 1800..2499: in no entry
    }
    entries = [("simplex", {"z": 1.0}), ("box", {"lower": 0.0, "upper": 0.7}), ("simplex_eq", {"z": 2.0})]
    col_proj = np.full(n, -1, dtype=np.int32)
    col_proj[0:1000], col_proj[1000:1800], col_proj[2500:4000] = 0, 1, 2
    for dn in ("f32", "f64"):
        f = _compare(p, pm, entries, col_proj, 0.03, dn, lam)
        # (column 1999 is in no entry: point-wise columns of any length are part of the window stream, not single-column tiles)
        assert f.info()["long_columns"] + f.info().get("slice_lane_columns", 0) >= 3  # (up to 512 non-zeros: K-lane slices of the second binary)


@pytest.mark.parametrize("host_pack", [False, True])
def test_pointwise_windows_cut_through_columns(host_pack, monkeypatch):
    """Box / cone / no-entry columns are streamed in windows of 256 non-zeros that ignore column boundaries (columns of 1 to 5000
    non-zeros, entries changing mid-window, empty columns): same numbers as the oracle, and -- the gradient being summed in
    integer fixed point -- the same BITS as a handle built with whole-column windows (DUALIP_HIP_FLAT=0)."""
    import os

    if os.environ.get("DUALIP_HIP_FLAT") == "0":
        pytest.skip("compares the flat windows with whole-column ones: needs the default")

    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.projections.base import ProjectionEntry

    if host_pack:
        monkeypatch.setenv("DUALIP_HIP_HOST_PACK", "1")
    m, n = 3_000, 5_000
    p = _random_problem(m, n, 12, seed=77, long_cols=((0, 700), (3, 256), (4, 1), (900, 3000), (901, 2999), (2400, 255), (2401, 257), (4200, 1023), (4999, 300)),
                        empty_every=9)
    lam = np.random.default_rng(12).uniform(0, 0.02, m)
    pm = {
        "box": ProjectionEntry("box", {"lower": 0.05, "upper": 0.6}, indices=list(range(0, 1500))),
        "cone": ProjectionEntry("cone", {"lower": 0.1}, indices=list(range(1500, 2400))),
        "simplex": ProjectionEntry("simplex", {"z": 1.0}, indices=list(range(2400, 3000))),
        "box2": ProjectionEntry("box", {"lower": 0.0, "upper": 0.3}, indices=list(range(3000, 4000))),
        # 4000..4999: in no entry
    }
    entries = [("box", {"lower": 0.05, "upper": 0.6}), ("cone", {"lower": 0.1}), ("simplex", {"z": 1.0}), ("box", {"lower": 0.0, "upper": 0.3})]
    col_proj = np.full(n, -1, dtype=np.int32)
    col_proj[0:1500], col_proj[1500:2400], col_proj[2400:3000], col_proj[3000:4000] = 0, 1, 2, 3
    for dn in ("f32", "f64"):
        f = _compare(p, pm, entries, col_proj, 0.04, dn, lam)
        td = torch.float32 if dn == "f32" else torch.float64
        lam_t = torch.from_numpy(lam).to(td).to(DEV)
        got = f.calculate(lam_t, save_primal=True)
        g1, x1 = got.dual_gradient.clone(), got.primal_var.clone()
        monkeypatch.setenv("DUALIP_HIP_FLAT", "0")
        whole = MatchingSolverDualObjectiveFunction(torch_args(p, dn, pm, DEV), gamma=0.04)
        monkeypatch.delenv("DUALIP_HIP_FLAT")
        assert whole.info()["long_columns"] > f.info()["long_columns"]
        ref = whole.calculate(lam_t, save_primal=True)
        assert torch.equal(ref.primal_var, x1)
        assert torch.equal(ref.dual_gradient, g1)
    # with the hot-rows plan (rows beyond the first 1024 gather / scatter through memory)
    monkeypatch.setenv("DUALIP_HIP_HOT_ROWS", "1024")
    f = _compare(p, pm, entries, col_proj, 0.04, "f32", lam)
    assert f.info()["hot_rows"] == 1024


def test_all_columns_empty():
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.projections import create_projection_map

    p = dict(m=50, n=30, colptr=np.zeros(31, dtype=np.int64), rowidx=np.zeros(0, dtype=np.int64), a=np.zeros(0), c=np.zeros(0), b=np.linspace(0.1, 1, 50))
    f = MatchingSolverDualObjectiveFunction(torch_args(p, "f64", create_projection_map("simplex", {"z": 1.0}, 30), DEV), gamma=0.1)
    lam = torch.rand(50, dtype=torch.float64, device=DEV)
    res = f.calculate(lam, save_primal=True)
    assert torch.allclose(res.dual_gradient, -torch.from_numpy(p["b"]).to(DEV))
    assert abs(float(res.dual_objective) + float((lam.cpu() * torch.from_numpy(p["b"])).sum())) < 1e-12 and res.primal_var.numel() == 0


@pytest.mark.parametrize("batching", [True, False])
def test_simplex_eq_reference_padding_with_long_columns(batching):
    """The padded-block mode through the single-column path (columns of 300-1500 non-zeros) and ordinary tiles, against the
    oracle's padded blocks (one oracle entry per nnz-bucket / one for the whole map)."""
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.projections import create_projection_map
    from tests.helpers import padded_eq_entries

    m, n = 2_000, 3_000
    p = _random_problem(m, n, 10, seed=33, long_cols=((1, 300), (17, 1500), (1999, 257)), empty_every=13)
    lam = np.random.default_rng(8).uniform(0, 0.01, m)
    zz = 400.0  # far above most clamped column sums: the padding decides the result
    entries, _, col_proj = padded_eq_entries(p, zz, batching)
    for dn in ("f32", "f64"):
        f = MatchingSolverDualObjectiveFunction(torch_args(p, dn, create_projection_map("simplex_eq", {"z": zz}, n), DEV), gamma=0.03, batching=batching,
                                                simplex_eq_padding="reference")
        td = torch.float32 if dn == "f32" else torch.float64
        res = f.calculate(torch.from_numpy(lam).to(td).to(DEV), save_primal=True)
        ax, obj0, ssq, x = oracle.matching_calculate(m, n, p["colptr"], p["rowidx"], p["a"], p["c"], lam, 0.03, entries, col_proj=col_proj, dtype=NP_DT[dn])
        grad = agd_oracle.epilogue(ax, obj0, ssq, lam, p["b"], 0.03, NP_DT[dn])[0]
        assert relerr(res.primal_var.cpu().numpy(), x) < RTOL[dn]
        assert relerr(res.dual_gradient.cpu().numpy(), grad) < RTOL[dn]


def test_results_are_bit_reproducible():
    """The gradient is accumulated in 64-bit fixed point (integer atomics are associative): repeated launches, and two
    independently built handles, return identical bits -- the reference's scatter_add_ on a GPU does not."""
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.projections.base import ProjectionEntry

    m, n = 500, 40_000
    p = _random_problem(m, n, 10, seed=77)
    pm = {
        "box": ProjectionEntry("box", {"lower": 0.0, "upper": 1.0}, indices=range(0, n // 2)),
        "simplex": ProjectionEntry("simplex", {"z": 1.0}, indices=range(n // 2, n)),
    }
    lam = torch.from_numpy(np.random.default_rng(3).uniform(0, 0.02, m)).float().to(DEV)
    outs = []
    for _ in range(2):
        f = MatchingSolverDualObjectiveFunction(torch_args(p, "f32", pm, DEV), gamma=0.02)
        for _ in range(3):
            r = f.calculate(lam, save_primal=True)
            outs.append((r.dual_gradient.clone(), r.primal_var.clone(), float(r.dual_objective)))
    for g, x, o in outs[1:]:
        assert torch.equal(g, outs[0][0]) and torch.equal(x, outs[0][1]) and o == outs[0][2]


def test_32_bit_gradient_slabs_are_the_same_exact_sums(monkeypatch):
    """fp32 handles with the whole gradient in LDS whose projections all bound x flush their per-workgroup gradient slabs as int32 LOW words
    of the 64-bit LDS accumulators (dl_matching_info 2007) -- and, only for a workgroup one of whose shares does not fit 32 bits, the high
    words too, stamped with the launch's epoch.  The fixed-point grid is taken from what a workgroup's share of a row is expected to stay
    below; nothing but the speed depends on that estimate.  Checked: bit-identical run to run and for ANY deal of the tiles (even / adapted);
    within fp32 rounding of the 64-bit-slab result and of the oracle; the overflow path (DUALIP_HIP_SLAB32=tiny: every workgroup overflows)
    gives the same sums on its finer grid; fp64 handles and one-sided projections keep 64-bit slabs; the honoured switches are reported."""
    import os

    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent
    from dualip_amd.projections import create_projection_map
    from dualip_amd.projections.base import ProjectionEntry

    if os.environ.get("DUALIP_HIP_SLAB32") is not None or os.environ.get("DUALIP_HIP_LDS_MODE") in ("grad", "none"):
        pytest.skip("states the default plan of the 256-wide layout")
    m, n = 500, 120_000  # (enough tiles for >= 128 workgroups: smaller handles keep 64-bit slabs -- a workgroup's share would be most of a row)
    p = _random_problem(m, n, 10, seed=77, long_cols=[(11, 300), (39_000, 90)])
    pm = {"box": ProjectionEntry("box", {"lower": 0.0, "upper": 1.0}, indices=range(0, n // 2)), "simplex": ProjectionEntry("simplex", {"z": 1.0}, indices=range(n // 2, n))}
    entries, col_proj = [("box", {"lower": 0.0, "upper": 1.0}), ("simplex", {"z": 1.0})], np.repeat(np.array([0, 1], dtype=np.int32), n // 2)
    lam_np = np.random.default_rng(3).uniform(0, 0.02, m)
    lam = torch.from_numpy(lam_np).float().to(DEV)
    kw = dict(max_iter=40, gamma=0.02, initial_step_size=1e-3, max_step_size=0.1, iteration_callback=False)

    def run(env):
        before = {k: os.environ.get(k) for k in env}
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        f = MatchingSolverDualObjectiveFunction(torch_args(p, "f32", pm, DEV), gamma=0.02)
        for k, v in before.items():  # (back to the ambient value: the suite is also run with one plan switch set for every test)
            monkeypatch.delenv(k) if v is None else monkeypatch.setenv(k, v)
        r = f.calculate(lam, save_primal=True)
        out = (r.dual_gradient.clone(), r.primal_var.clone(), float(r.dual_objective), f.info())
        res = AcceleratedGradientDescent(**kw).maximize(f, torch.zeros(m, dtype=torch.float32, device=DEV))
        return out + (list(res.dual_objective_log), res.dual_val.clone(), f.info()["slab_overflows"])

    ambient = sorted(k for k in os.environ if k.startswith("DUALIP_HIP_"))  # (the suite is also run once per plan switch: tools/suite_under_switches.sh)
    g32, x32, o32, info32, log32, d32, _ = run({})
    assert info32["slab_bytes"] == 4 and info32["workgroups"] >= 128 and sorted(info32["switches"]) == ambient and info32["developer_build"] == 0 and info32["slab_overflows"] == 0, info32
    g32b, x32b, o32b, info_b, log32b, d32b, _ = run({"DUALIP_HIP_XCD_BALANCE": "0", "DUALIP_HIP_SELL_BALANCE": "0"})  # another deal, the same integers
    assert sorted(info_b["switches"]) == sorted(set(ambient) | {"DUALIP_HIP_SELL_BALANCE", "DUALIP_HIP_XCD_BALANCE"})
    assert torch.equal(g32, g32b) and torch.equal(x32, x32b) and o32 == o32b and log32 == log32b and torch.equal(d32, d32b)
    g64, x64, o64, info64, log64, d64, _ = run({"DUALIP_HIP_SLAB32": "0"})
    assert info64["slab_bytes"] == 8 and sorted(info64["switches"]) == sorted(set(ambient) | {"DUALIP_HIP_SLAB32"})
    assert torch.equal(x32, x64)  # (the primal does not pass through the slabs)
    assert relerr(g32.cpu().numpy(), g64.cpu().numpy()) < 2e-7 and abs(o32 - o64) <= 1e-6 * abs(o64)
    assert relerr(log32, log64) < 1e-6 and relerr(d32.cpu().numpy(), d64.cpu().numpy()) < 1e-5
    # every workgroup on the overflow path: low AND high words travel, the sums are those of a (finer) grid
    gt, xt, ot, info_t, logt, dt, ovf_t = run({"DUALIP_HIP_SLAB32": "tiny"})
    assert info_t["slab_bytes"] == 4 and info_t["slab_overflows"] > info_t["workgroups"] // 2 and ovf_t > info_t["workgroups"] // 2, info_t
    assert torch.equal(xt, x64) and relerr(gt.cpu().numpy(), g64.cpu().numpy()) < 2e-7 and relerr(logt, log64) < 1e-6
    ax, obj0, ssq, x = oracle.matching_calculate(m, n, p["colptr"], p["rowidx"], p["a"], p["c"], lam_np, 0.02, entries, col_proj=col_proj, dtype=np.float32)
    grad, obj, *_ = agd_oracle.epilogue(ax, obj0, ssq, lam_np, p["b"], 0.02, np.float32)
    assert relerr(g32.cpu().numpy(), grad) < RTOL["f32"] and relerr([o32], [obj]) < RTOL["f32"] * 10
    # the developer switches do not exist in the shipped library: an ablation request changes nothing
    monkeypatch.setenv("DUALIP_HIP_ABLATE", "7")
    ga, xa, oa, info_a, *_ = run({})
    monkeypatch.delenv("DUALIP_HIP_ABLATE")
    assert torch.equal(ga, g32) and torch.equal(xa, x32) and sorted(info_a["switches"]) == ambient
    # who keeps 64-bit slabs: fp64 handles, maps with a one-sided operator
    assert MatchingSolverDualObjectiveFunction(torch_args(p, "f64", pm, DEV), gamma=0.02).info()["slab_bytes"] == 8
    assert MatchingSolverDualObjectiveFunction(torch_args(p, "f32", create_projection_map("cone", {"lower": 0.0}, n), DEV), gamma=0.02).info()["slab_bytes"] == 8
    q = _random_problem(300, 20_000, 8, seed=5)  # a handle that does not fill the chip
    small = MatchingSolverDualObjectiveFunction(torch_args(q, "f32", create_projection_map("box", {"lower": 0.0, "upper": 1.0}, q["n"]), DEV), gamma=0.02).info()
    assert small["workgroups"] < 128 and small["slab_bytes"] == 8, small


def test_32_bit_slabs_are_refused_when_row_scales_differ(monkeypatch):
    """The grid of the 32-bit slabs is ONE value for the whole matrix, taken from the largest row L1 norm: a row orders of magnitude smaller
    would collect rounding noise that is large against its own sum (every a x is rounded to the grid before the integer add).  Handles whose
    rows fail  L1_i / sqrt(count_i) >= kSlabNoise * slab_abound  (api.hip: slab_refresh_bound) therefore keep the 64-bit slabs: here rows
    scaled 1e-3 .. 1e3 (beyond ~1e7 of spread the 2^50 grid of the 64-bit slabs is itself no longer at fp32 level for the smallest rows).
    Checked PER ROW -- relative to the row's own sum of |a x|, not to the largest gradient entry -- against the float64 sum of the handle's
    own fp32 products: the default plan stays within fp32 accumulation error on every row; the forced 32-bit grid (DUALIP_HIP_SLAB32=force)
    does not, which is why it is refused.  The same pattern with unit row L1 norms (what Jacobi preconditioning produces) passes the gate."""
    import os

    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.projections.base import ProjectionEntry

    if os.environ.get("DUALIP_HIP_SLAB32") is not None or os.environ.get("DUALIP_HIP_LDS_MODE") in ("grad", "none"):
        pytest.skip("states the default plan of the 256-wide layout")
    m, n = 500, 120_000
    p = _random_problem(m, n, 10, seed=78)
    scale = 10.0 ** np.linspace(-3, 3, m)
    np.random.default_rng(5).shuffle(scale)
    p["a"] = p["a"] * scale[p["rowidx"]]
    pm = {"box": ProjectionEntry("box", {"lower": 0.0, "upper": 1.0}, indices=range(0, n // 2)), "simplex": ProjectionEntry("simplex", {"z": 1.0}, indices=range(n // 2, n))}
    lam = torch.from_numpy(np.random.default_rng(3).uniform(0, 0.02, m) / scale).float().to(DEV)  # (duals of rows in other units scale inversely: a lambda stays O(c))
    gamma = 0.02
    a32 = p["a"].astype(np.float32)

    def row_errors(env=None):
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        f = MatchingSolverDualObjectiveFunction(torch_args(p, "f32", pm, DEV), gamma=gamma)
        for k in (env or {}):
            monkeypatch.delenv(k)
        ax = f.calculate_packed(lam, gamma, x_out=f._primal_buffer()).clone()[:m].cpu().numpy()  # A x as the slabs summed it (float64 of the integers)
        x = f._primal_buffer().clone().cpu().numpy()
        prod = (a32 * x).astype(np.float64)  # the fp32 products the kernel forms, added up in float64
        want, mag = np.zeros(m), np.zeros(m)
        np.add.at(want, p["rowidx"], prod)
        np.add.at(mag, p["rowidx"], np.abs(prod))
        live = mag > 0
        return np.abs(ax - want)[live] / mag[live], scale[live], x, f.info()

    err, sc, x_d, info_d = row_errors()
    assert info_d["slab_bytes"] == 8 and info_d["slab_rows_ok"] == 0 and info_d["workgroups"] >= 128, info_d  # refused: 64-bit slabs
    assert err.max() < 1e-6, err.max()  # every row within fp32-level error of ITS OWN magnitude
    errf, scf, x_f, info_f = row_errors({"DUALIP_HIP_SLAB32": "force"})
    assert info_f["slab_bytes"] == 4 and info_f["slab_rows_ok"] == 0 and np.array_equal(x_f, x_d), info_f
    assert errf.max() > 1e-4 and errf.max() > 50 * err.max(), (errf.max(), err.max())  # the small rows on the forced grid: what the gate prevents
    assert errf[scf >= 1.0].max() < 1e-6  # (the large rows are fine on either grid)
    l1 = np.zeros(m)
    np.add.at(l1, p["rowidx"], np.abs(p["a"]))
    p["a"] = p["a"] / l1[p["rowidx"]]
    ok = MatchingSolverDualObjectiveFunction(torch_args(p, "f32", pm, DEV), gamma=gamma).info()
    assert ok["slab_bytes"] == 4 and ok["slab_rows_ok"] == 1, ok


def test_32_bit_slabs_on_the_finest_rows_grid_with_wide_rows(monkeypatch):
    """Row scales that differ by orders of magnitude but with only a FEW large rows: the grid of the 32-bit slabs is taken from the row that needs
    the finest one (every row's rounding noise below 2^-20 of its own L1 norm), and the few rows whose workgroup shares do not fit 32 bits on that
    grid are WIDE -- every workgroup sends their high words in every launch (api.hip: slab_refresh_bound).  Here: 10 rows at scale 1, 480 at 1e-3,
    10 at 1e-6.  Checked: 32-bit slabs are on with the ten large rows wide and no dynamic overflow; EVERY row -- the 1e-6 ones included -- is within
    fp32-level error of its own magnitude (round 5's grid, DUALIP_HIP_SLAB32=force, is not); the sums are bit-identical for another deal of the
    tiles and within fp32 rounding of the 64-bit slabs; a solve logs the same numbers under both deals."""
    import os

    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent
    from dualip_amd.projections.base import ProjectionEntry

    if os.environ.get("DUALIP_HIP_SLAB32") is not None or os.environ.get("DUALIP_HIP_LDS_MODE") in ("grad", "none"):
        pytest.skip("states the default plan of the 256-wide layout")
    m, n = 500, 120_000
    p = _random_problem(m, n, 10, seed=79)
    scale = np.full(m, 1e-3)
    scale[:10] = 1.0
    scale[10:20] = 1e-6
    np.random.default_rng(6).shuffle(scale)
    p["a"] = p["a"] * scale[p["rowidx"]]
    p["b"] = p["b"] * scale * 100.0
    pm = {"box": ProjectionEntry("box", {"lower": 0.0, "upper": 1.0}, indices=range(0, n // 2)), "simplex": ProjectionEntry("simplex", {"z": 1.0}, indices=range(n // 2, n))}
    lam = torch.from_numpy(np.random.default_rng(3).uniform(0, 0.02, m) / scale).float().to(DEV)
    gamma = 0.02
    a32 = p["a"].astype(np.float32)
    kw = dict(max_iter=30, gamma=gamma, initial_step_size=1e-6, max_step_size=1e-4, iteration_callback=False)

    def run(env=None):
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        f = MatchingSolverDualObjectiveFunction(torch_args(p, "f32", pm, DEV), gamma=gamma)
        for k in (env or {}):
            monkeypatch.delenv(k)
        ax = f.calculate_packed(lam, gamma, x_out=f._primal_buffer()).clone()[:m].cpu().numpy()
        x = f._primal_buffer().clone().cpu().numpy()
        info = f.info()
        prod = (a32 * x).astype(np.float64)
        want, mag = np.zeros(m), np.zeros(m)
        np.add.at(want, p["rowidx"], prod)
        np.add.at(mag, p["rowidx"], np.abs(prod))
        live = mag > 0
        res = AcceleratedGradientDescent(**kw).maximize(f, torch.zeros(m, dtype=torch.float32, device=DEV))
        return ax, np.abs(ax - want)[live] / mag[live], scale[live], x, info, list(res.dual_objective_log), res.dual_val.clone()

    ax, err, sc, x, info, log, dual = run()
    assert info["slab_bytes"] == 4 and info["slab_rows_ok"] == 1 and info["slab_wide_rows"] == 10 and info["slab_overflows"] == 0, info
    # every row, the tiny ones included: the criterion is noise <= 2^-20 of L1_i * xmax; relative to the row's sum of |a x| (x well below its bound) a few times that
    assert err.max() < 2e-5 and err[sc >= 1e-3].max() < 1e-6, (err.max(), err[sc >= 1e-3].max())
    ax_b, err_b, _, x_b, info_b, log_b, dual_b = run({"DUALIP_HIP_XCD_BALANCE": "0", "DUALIP_HIP_SELL_BALANCE": "0"})  # another deal: the same integers
    assert np.array_equal(ax, ax_b) and np.array_equal(x, x_b) and log == log_b and torch.equal(dual, dual_b) and info_b["slab_wide_rows"] == 10
    ax64, err64, _, x64, info64, log64, dual64 = run({"DUALIP_HIP_SLAB32": "0"})
    assert info64["slab_bytes"] == 8 and np.array_equal(x, x64)
    assert relerr(ax, ax64) < 2e-7 and relerr(log, log64) < 1e-6
    # values_changed(): the grid and the list of wide rows follow the rows' norms -- a handle built on homogeneous rows (no wide row), its values
    # rescaled in place to this problem's, must return what a fresh handle returns; and one more rescaling (spread 1e6) sends it to 64-bit slabs
    q = dict(p, a=p["a"] / scale[p["rowidx"]])
    args_q = torch_args(q, "f32", pm, DEV)
    fq = MatchingSolverDualObjectiveFunction(args_q, gamma=gamma)
    assert fq.info()["slab_bytes"] == 4 and fq.info()["slab_wide_rows"] == 0
    rs = torch.from_numpy(scale).float().to(DEV)[args_q.A.row_indices()]
    args_q.A.values().mul_(rs)
    fq.values_changed()
    assert fq.info()["slab_bytes"] == 4 and fq.info()["slab_wide_rows"] == 10, fq.info()
    fresh = MatchingSolverDualObjectiveFunction(torch_args(dict(p, a=args_q.A.values().double().cpu().numpy()), "f32", pm, DEV), gamma=gamma)
    assert fresh.info()["slab_wide_rows"] == 10 and torch.equal(fq.calculate_packed(lam, gamma)[:m], fresh.calculate_packed(lam, gamma)[:m])
    assert relerr(fq.calculate_packed(lam, gamma)[:m].cpu().numpy(), ax) < 1e-6  # (the rescaled values are this problem's up to an fp32 rounding)
    spread = torch.from_numpy(10.0 ** np.linspace(-3, 3, m)).float().to(DEV)[args_q.A.row_indices()]
    args_q.A.values().mul_(spread)
    fq.values_changed()
    assert fq.info()["slab_bytes"] == 8 and fq.info()["slab_rows_ok"] == 0, fq.info()
    fresh = MatchingSolverDualObjectiveFunction(torch_args(dict(p, a=args_q.A.values().double().cpu().numpy()), "f32", pm, DEV), gamma=gamma)
    assert torch.equal(fq.calculate_packed(lam, gamma)[:m], fresh.calculate_packed(lam, gamma)[:m])
    axf, errf, scf, xf, info_f, *_ = run({"DUALIP_HIP_SLAB32": "force"})  # round 5's grid, from the largest row
    assert info_f["slab_bytes"] == 4 and info_f["slab_wide_rows"] == 0 and info_f["slab_rows_ok"] == 0
    assert errf[scf < 1e-4].max() > 5e-4 and errf[scf < 1e-4].max() > 20 * err[sc < 1e-4].max() and errf[scf >= 1.0].max() < 1e-6, (errf[scf < 1e-4].max(), err[sc < 1e-4].max())


def _skewed_problem(m, n, mean_deg, seed):
    """Rows drawn from a heavy-tailed popularity law (a few destinations get most of the edges)."""
    rng = np.random.default_rng(seed)
    w = rng.lognormal(0.0, 1.5, m)
    w /= w.sum()
    deg = np.minimum(rng.poisson(mean_deg, n), 64).astype(np.int64)
    colptr = np.zeros(n + 1, dtype=np.int64)
    rows = []
    for j in range(n):
        r = np.unique(rng.choice(m, size=int(deg[j]), p=w)) if deg[j] else np.zeros(0, dtype=np.int64)
        rows.append(r)
        colptr[j + 1] = colptr[j] + r.size
    rows = np.concatenate(rows).astype(np.int64)
    nnz = rows.size
    return dict(m=m, n=n, colptr=colptr, rowidx=rows, a=rng.uniform(0.05, 1.0, nnz), c=-rng.uniform(0.01, 0.5, nnz), b=rng.uniform(0.5, 2.0, m))


@pytest.mark.parametrize("forced", [True, False, "whole_dual_vector", "whole_dual_vector_off"])
def test_hot_rows_plan(forced, monkeypatch):
    """Dual vector + gradient larger than the LDS: rows renumbered by frequency, the hot ones in LDS, the cold tail on L2
    gathers / global atomics.  Natural case: 30 000 dual rows; forced case: a small problem with only 128 hot rows.
    Checked: one calculate() (slab reduction with the inverse permutation) and a device-resident AGD run (stats kernel
    reading the renumbered slabs) against the oracle."""
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent
    from dualip_amd.projections.base import ProjectionEntry

    # (round 4) "whole_dual_vector": 20 000 rows -- in fp32 the WHOLE dual vector fits the LDS beside ~9 000 gradient rows, so the plan
    # stages all of it (no tile gathers from L2; only the scatter of the rows beyond the gradient's share leaves the CU); DUALIP_HIP_LAM_ALL=0
    # keeps the symmetric plan on the same problem
    whole = isinstance(forced, str)
    if whole:
        if forced.endswith("_off"):
            monkeypatch.setenv("DUALIP_HIP_LAM_ALL", "0")
        p = _skewed_problem(20_000, 8_000, 12, seed=43)
        forced_off, forced = forced.endswith("_off"), False
    elif forced:
        monkeypatch.setenv("DUALIP_HIP_HOT_ROWS", "128")
        p = _skewed_problem(700, 5_000, 9, seed=41)
    else:
        p = _skewed_problem(30_000, 8_000, 12, seed=42)
    m, n = p["m"], p["n"]
    pm = {
        "box": ProjectionEntry("box", {"lower": 0.0, "upper": 1.0}, indices=list(range(0, n // 2))),
        "simplex": ProjectionEntry("simplex", {"z": 1.0}, indices=list(range(n // 2, n))),
    }
    entries = [("box", {"lower": 0.0, "upper": 1.0}), ("simplex", {"z": 1.0})]
    col_proj = np.zeros(n, dtype=np.int32)
    col_proj[n // 2 :] = 1
    lam = np.random.default_rng(5).uniform(0, 0.02, m)
    import os

    narrow = False  # (one tile layout since round 5)
    for dn in ("f32", "f64"):
        f = _compare(p, pm, entries, col_proj, 0.05, dn, lam)
        info = f.info()
        if narrow:
            assert info["hot_rows"] == 0
            continue
        assert info["hot_rows"] == (128 if forced else info["hot_rows"]) and 0 < info["hot_rows"] < m, info
        assert info["lambda_in_lds"] == 1 and info["grad_in_lds"] == 1
        # the per-XCD cold-row accumulators are taken only after the device passed their self-check (2048 contending workgroups, exact
        # counts): on this part it does; DUALIP_HIP_COLD_XCD=0 (or a failed check) selects the shared array
        assert info["cold_per_xcd"] == (0 if os.environ.get("DUALIP_HIP_COLD_XCD") == "0" else 1), info
        if whole and dn == "f32" and not forced_off:
            assert info["lambda_rows_in_lds"] == m and 1024 <= info["hot_rows"] < 12_000, info
        else:
            assert info["lambda_rows_in_lds"] == info["hot_rows"], info
        if not forced:
            assert info["hot_nnz_ppm"] > 500_000  # the frequent rows carry most of the non-zeros
    # device-resident AGD over the renumbered slabs
    f = MatchingSolverDualObjectiveFunction(torch_args(p, "f64", pm, DEV), gamma=0.05)
    solver = AcceleratedGradientDescent(max_iter=40, gamma=0.05, initial_step_size=1e-4, max_step_size=1e-2, iteration_callback=False)
    res = solver.maximize(f, torch.zeros(m, dtype=torch.float64, device=DEV))

    def calc(lam_, gamma):
        ax, obj0, ssq, _ = oracle.matching_calculate(m, n, p["colptr"], p["rowidx"], p["a"], p["c"], lam_, gamma, entries, col_proj=col_proj, dtype=np.float64, want_x=False)
        grad, obj, *_ = agd_oracle.epilogue(ax, obj0, ssq, lam_, p["b"], gamma, np.float64)
        return grad, obj, None

    want = agd_oracle.maximize(calc, np.zeros(m), 40, 0.05, initial_step_size=1e-4, max_step_size=1e-2, dtype=np.float64)
    assert relerr(res.dual_objective_log, want["dual_obj_log"]) < 1e-8
    assert relerr(res.dual_val.cpu().numpy(), want["dual_val"]) < 1e-8


def test_hot_rows_with_single_column_tiles_and_primal(monkeypatch):
    """Hot-rows plan together with columns longer than a window (their walker gathers / scatters cold rows too) and the
    primal written out."""
    from dualip_amd.projections import create_projection_map

    monkeypatch.setenv("DUALIP_HIP_HOT_ROWS", "192")
    monkeypatch.setenv("DUALIP_HIP_FLAT", "0")  # the single-column walker is the subject: keep long point-wise columns out of the window stream
    p = _random_problem(900, 3_000, 10, seed=55, long_cols=((3, 400), (1500, 700), (2999, 260)), empty_every=17)
    lam = np.random.default_rng(9).uniform(0, 0.02, p["m"])
    for dn in ("f32", "f64"):
        for pt, pp in (("simplex", {"z": 1.0}), ("box", {"lower": 0.0, "upper": 0.5})):
            entries = [(pt, pp)]
            f = _compare(p, create_projection_map(pt, dict(pp), p["n"]), entries, None, 0.05, dn, lam)
            info = f.info()
            if info["layout"] == 4:
                assert info["hot_rows"] == 192 and info["long_columns"] + info.get("slice_lane_columns", 0) >= 3


@pytest.mark.parametrize("forced", [False, True])
def test_columns_walked_by_a_whole_workgroup(forced, monkeypatch):
    """Columns of thousands of non-zeros are walked by all 16 wavefronts of a workgroup together (one wavefront alone would
    set the critical path of the launch); ``forced`` lowers the threshold so that every single-column tile takes that path.
    Simplex (several Newton passes over a 9 000-entry support), simplex_eq exact and padded, point-wise, columns in no
    entry, with and without the hot-rows plan, primal written out."""
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.projections import create_projection_map
    from dualip_amd.projections.base import ProjectionEntry
    from tests.helpers import padded_eq_entries

    if forced:
        monkeypatch.setenv("DUALIP_HIP_XLONG_MIN", "256")
    monkeypatch.setenv("DUALIP_HIP_FLAT", "0")  # the walkers are the subject: keep long point-wise columns out of the window stream
    m, n = 10_000, 3_000
    # (round 4: the second binary walks simplex columns of up to 2048 non-zeros by ONE wavefront as in-place slices of 20 / 24 / 28 / 32 steps
    #  -- 1100, 1500, 1700, 2048 below -- so only the columns beyond 2048 go to the whole workgroup there; 2048 | 2049 is the class edge)
    long_cols = ((2, 9000), (700, 3000), (1500, 2049), (2999, 2048), (5, 600), (2000, 5000), (100, 1100), (200, 1500), (300, 1700))
    p = _random_problem(m, n, 10, seed=77, long_cols=long_cols, empty_every=19)
    lam = np.random.default_rng(10).uniform(0, 0.01, m)

    def want_of(info):  # columns the whole workgroup walks: all 9 single-column tiles when forced (threshold 256), else those beyond 1024 / 2048
        return 9 if forced else (4 if info["second_binary"] else 8)
    for dn in ("f32", "f64"):
        for pt, pp in (("simplex", {"z": 1.0}), ("simplex", {"z": 40.0}), ("box", {"lower": 0.0, "upper": 0.5})):
            f = _compare(p, create_projection_map(pt, dict(pp), n), [(pt, pp)], None, 0.05, dn, lam)
            info = f.info()
            assert info["long_columns"] >= 9 and info["workgroup_columns"] == want_of(info), info
        # simplex_eq, exact mode (the oracle's padded blocks differ wherever a clamped column sums to less than z): every
        # non-empty column sums to z, and columns without a deficit agree with the oracle
        td = torch.float32 if dn == "f32" else torch.float64
        f = MatchingSolverDualObjectiveFunction(torch_args(p, dn, create_projection_map("simplex_eq", {"z": 25.0}, n), DEV), gamma=0.05)
        x = f.calculate(torch.from_numpy(lam).to(td).to(DEV), save_primal=True).primal_var.cpu().numpy().astype(np.float64)
        _, _, _, xo = oracle.matching_calculate(m, n, p["colptr"], p["rowidx"], p["a"], p["c"], lam, 0.05, [("simplex_eq", {"z": 25.0})], dtype=NP_DT[dn])
        v = p["a"] * (-lam / 0.05)[p["rowidx"]] - p["c"] / 0.05
        assert x.min() >= 0
        for j in range(n):
            k0, k1 = int(p["colptr"][j]), int(p["colptr"][j + 1])
            if k1 > k0:
                assert abs(x[k0:k1].sum() - 25.0) < 25.0 * (1e-4 if dn == "f32" else 1e-10), j
                if np.maximum(v[k0:k1], 0).sum() > 25.5:
                    assert np.abs(x[k0:k1] - xo[k0:k1]).max() < RTOL[dn] * 25, j
        pm = {
            "s": ProjectionEntry("simplex", {"z": 3.0}, indices=list(range(0, 1000))),
            "b": ProjectionEntry("box", {"lower": 0.0, "upper": 0.7}, indices=list(range(1000, 1800))),
            "e": ProjectionEntry("simplex_eq", {"z": 2.0}, indices=list(range(2500, 3000))),
        }
        col_proj = np.full(n, -1, dtype=np.int32)
        col_proj[0:1000], col_proj[1000:1800], col_proj[2500:3000] = 0, 1, 2
        _compare(p, pm, [("simplex", {"z": 3.0}), ("box", {"lower": 0.0, "upper": 0.7}), ("simplex_eq", {"z": 2.0})], col_proj, 0.03, dn, lam)
    # padded simplex_eq blocks through the workgroup walker
    zz = 4000.0
    entries, _, col_proj = padded_eq_entries(p, zz, True)
    f = MatchingSolverDualObjectiveFunction(torch_args(p, "f64", create_projection_map("simplex_eq", {"z": zz}, n), DEV), gamma=0.03, simplex_eq_padding="reference")
    res = f.calculate(torch.from_numpy(lam).to(DEV), save_primal=True)
    ax, obj0, ssq, x = oracle.matching_calculate(m, n, p["colptr"], p["rowidx"], p["a"], p["c"], lam, 0.03, entries, col_proj=col_proj, dtype=np.float64)
    assert relerr(res.primal_var.cpu().numpy(), x) < RTOL["f64"]
    # hot-rows plan: the walker gathers / scatters cold rows through memory
    monkeypatch.setenv("DUALIP_HIP_HOT_ROWS", "2048")
    for dn in ("f32", "f64"):
        f = _compare(p, create_projection_map("simplex", {"z": 1.0}, n), [("simplex", {"z": 1.0})], None, 0.05, dn, lam)
        assert f.info()["layout"] != 4 or (f.info()["hot_rows"] == 2048 and f.info()["workgroup_columns"] == want_of(f.info()))


def test_hot_rows_state_survives_outside_calls_between_iterations(monkeypatch):
    """The device-resident loop leaves the renumbered dual vector and the zeroed cold accumulators ready for its next fused
    launch; a calculate() from outside on the same objective (here: from the iteration callback, with another dual vector)
    must not be mistaken for that state, and neither must a second run on the same objective."""
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent
    from dualip_amd.projections import create_projection_map

    monkeypatch.setenv("DUALIP_HIP_HOT_ROWS", "128")
    p = _skewed_problem(700, 5_000, 9, seed=43)
    m = p["m"]
    f = MatchingSolverDualObjectiveFunction(torch_args(p, "f64", create_projection_map("simplex", {"z": 1.0}, p["n"]), DEV), gamma=0.05)
    if f.info()["layout"] == 4:
        assert f.info()["hot_rows"] == 128
    kw = dict(max_iter=30, gamma=0.05, initial_step_size=1e-4, max_step_size=1e-2)
    zero = torch.zeros(m, dtype=torch.float64, device=DEV)
    plain = AcceleratedGradientDescent(iteration_callback=False, **kw).maximize(f, zero)
    other = torch.rand(m, dtype=torch.float64, device=DEV) * 0.05
    seen = []

    def intrude(it, result):
        seen.append(float(f.calculate(other).dual_objective))

    poked = AcceleratedGradientDescent(iteration_callback=intrude, **kw).maximize(f, zero)
    again = AcceleratedGradientDescent(iteration_callback=False, **kw).maximize(f, zero)
    assert len(seen) == 30 and max(seen) - min(seen) == 0.0
    assert poked.dual_objective_log == plain.dual_objective_log and again.dual_objective_log == plain.dual_objective_log
    assert torch.equal(poked.dual_val, plain.dual_val) and torch.equal(again.dual_val, plain.dual_val)
    # a warm start from another vector on the same objective
    warm = AcceleratedGradientDescent(iteration_callback=False, **kw).maximize(f, other)
    ref = MatchingSolverDualObjectiveFunction(torch_args(p, "f64", create_projection_map("simplex", {"z": 1.0}, p["n"]), DEV), gamma=0.05)
    assert AcceleratedGradientDescent(iteration_callback=False, **kw).maximize(ref, other).dual_objective_log == warm.dual_objective_log


def test_handles_on_two_devices_in_one_process():
    """The opt-in to more than 64 KB of LDS is a per-device function attribute: a process that builds handles on two devices (the
    reference's split_tensors_to_devices idiom) must get it on both (a per-process flag gave it to the first only)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.projections import create_projection_map
    from tests.helpers import load, problem, torch_args

    z = load("g1_syn2000.npz")
    p = problem(z)
    pm = create_projection_map("simplex", {"z": 1.0}, p["n"])
    out = []
    for dev in ("cuda:0", "cuda:1"):
        f = MatchingSolverDualObjectiveFunction(torch_args(p, "f64", pm, dev), 0.02)
        out.append(f.calculate(torch.from_numpy(z["lam_small"]).to(dev), 0.02).dual_gradient.cpu())
    assert torch.equal(out[0], out[1])


def test_xcd_weighted_deal_keeps_every_tile(monkeypatch):
    """The cyclic deal of window tiles to wavefronts with a per-workgroup number of rounds (csrc/fused_common.h: Deal), adapted from
    the launches' stamps.  Only large problems adapt by default; here the threshold is lowered so that a 3M-entity mixed problem does.
    Whatever table the timings produce, every tile keeps exactly one slot: the gradient (integer fixed point) and the primal
    are bit-identical to a handle with the even deal, the floating-point objective sums agree to rounding."""
    import os

    from benchmark.synthetic import generate_matching_problem
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.projections import create_projection_map

    n, m = 3_000_000, 2_000
    prob = generate_matching_problem(n, m, 5e-3, seed=5, device=torch.device(DEV), dtype=torch.float32)
    inp = prob["input_args"]
    half = n // 2
    inp.projection_map = {**create_projection_map("box", {"lower": 0.0, "upper": 1.0}, n, indices=range(0, half)),
                          **create_projection_map("simplex", {"z": 1.0}, n, indices=range(half, n))}
    lam = torch.rand(m, device=DEV) * 0.01
    monkeypatch.setenv("DUALIP_HIP_XCD_BALANCE", "0")
    even = MatchingSolverDualObjectiveFunction(inp, 1e-2)
    assert even._lib.dl_matching_info(even._handle, 18) == -1
    want = even.calculate(lam, save_primal=True)
    wg, wx, wo = want.dual_gradient.clone(), want.primal_var.clone(), float(want.dual_objective)
    monkeypatch.setenv("DUALIP_HIP_XCD_BALANCE", "1")
    monkeypatch.setenv("DUALIP_HIP_XCD_BALANCE_MIN_ROUNDS", "4")
    f = MatchingSolverDualObjectiveFunction(inp, 1e-2)
    info = f.info()
    n_wg = info["workgroups"]
    tables = set()
    for it in range(12):  # the first 8 launches adapt the table
        got = f.calculate(lam, save_primal=(it % 3 == 0))
        tab = tuple(int(f._lib.dl_matching_info(f._handle, 18 + i)) for i in range(n_wg))
        tables.add(tab)
        assert min(tab) >= 1 and max(tab) - min(tab) <= 64
        assert sum(tab) * 16 >= info["tiles"] - info["long_columns"]
        assert torch.equal(got.dual_gradient, wg)
        if it % 3 == 0:
            assert torch.equal(got.primal_var, wx)
        assert abs(float(got.dual_objective) - wo) <= 1e-6 * abs(wo)
    # (the timings of a real device are never perfectly even: the table moves; if it ever did not, the test still passed the invariants)
    print("tables seen", len(tables))


def test_two_phase_deal_of_the_slices_keeps_every_slice(monkeypatch):
    """The one-lane slices of a handle whose window tiles do not adapt (an all-simplex map) are dealt in two phases (csrc/fused4_kernel.h):
    everything up to n1 to every wavefront, the table's tail only to the workgroups that have been finishing early (csrc/matching_kernels.hip:
    sell_balance_kernel moves the cut and the membership from the launches' stamps).  Whatever the table, every slice keeps exactly one
    slot: gradient (integer fixed point), primal and -- the slices add one rounded integer per lane -- the objective are bit-identical to a
    handle with the even deal.  Fixed shares (the test switch) of 0.1 %, 5 %, 33 % and 90 %, then the adapting table at a lowered threshold."""
    import os

    from benchmark.synthetic import generate_matching_problem
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.projections import create_projection_map

    if os.environ.get("DUALIP_HIP_SELL") == "0":
        pytest.skip("no column-per-lane slices under this switch")
    if os.environ.get("DUALIP_HIP_XCD_BALANCE") == "0":
        pytest.skip("DUALIP_HIP_XCD_BALANCE=0 switches every adaptive deal off, the slices' two-phase one included")
    n, m = 1_500_000, 2_000
    prob = generate_matching_problem(n, m, 5e-3, seed=9, device=torch.device(DEV), dtype=torch.float32)
    inp = prob["input_args"]
    inp.projection_map = create_projection_map("simplex", {"z": 1.0}, n, indices=range(n))
    lam = torch.rand(m, device=DEV) * 0.01
    monkeypatch.setenv("DUALIP_HIP_SELL_BALANCE", "0")
    even = MatchingSolverDualObjectiveFunction(inp, 1e-2)
    assert even.info()["slice_balance_ppm"] == -1
    want = even.calculate(lam, save_primal=True)
    wg, wx, wo = want.dual_gradient.clone(), want.primal_var.clone(), float(want.dual_objective)
    monkeypatch.delenv("DUALIP_HIP_SELL_BALANCE")
    for ppm in (1000, 50_000, 330_000, 900_000):
        monkeypatch.setenv("DUALIP_HIP_SELL_BALANCE_PPM", str(ppm))
        f = MatchingSolverDualObjectiveFunction(inp, 1e-2)
        info = f.info()
        if info["second_binary"]:
            pytest.skip("this handle takes the second binary (its own dynamic deal)")
        assert info["slice_balance_ppm"] == ppm
        for it in range(3):
            got = f.calculate(lam, save_primal=(it == 0))
            assert torch.equal(got.dual_gradient, wg), ppm
            if it == 0:
                assert torch.equal(got.primal_var, wx), ppm
            assert float(got.dual_objective) == wo, ppm
        assert f.info()["slice_balance_updates"] == 0  # (a fixed table is never adapted)
    monkeypatch.delenv("DUALIP_HIP_SELL_BALANCE_PPM")
    monkeypatch.setenv("DUALIP_HIP_XCD_BALANCE_MIN_ROUNDS", "4")
    f = MatchingSolverDualObjectiveFunction(inp, 1e-2)
    assert f.info()["slice_balance_ppm"] == 0
    shares = set()
    for it in range(12):  # the first 8 launches adapt
        got = f.calculate(lam, save_primal=(it % 4 == 0))
        assert torch.equal(got.dual_gradient, wg)
        if it % 4 == 0:
            assert torch.equal(got.primal_var, wx)
        assert float(got.dual_objective) == wo
        i2 = f.info()
        assert 0 <= i2["slice_balance_ppm"] <= 200_000
        shares.add(i2["slice_balance_ppm"])
    assert f.info()["slice_balance_updates"] >= 8
    print("shares seen (ppm)", sorted(shares))


def test_non_finite_values_are_refused_and_rescaled_costs_keep_the_objective_exact():
    """The gradient, c.x and sum x^2 are exact fixed-point sums whose grids come from max |a|, max |c| (fused_common.h).  A value array
    with an inf or NaN has no such grid: the handle refuses it (the reference would return NaN) instead of logging integer garbage.
    And costs rewritten in place -- here scaled by 10^7 under a BOUNDED map -- must move the grid of the scalar sums with them:
    ``costs_changed()`` refreshes max |c| for every map (round 3 did so only for unbounded projections)."""
    from dualip_amd._hip import HipLibraryError
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.projections import create_projection_map

    p = _random_problem(300, 5000, 8, seed=5)
    pm = create_projection_map("box", {"lower": 0.0, "upper": 1.0}, p["n"])
    for which, bad in (("a", np.nan), ("c", np.inf)):
        q = dict(p)
        q[which] = p[which].copy()
        q[which][1234] = bad
        with pytest.raises((HipLibraryError, RuntimeError, ValueError), match="inf or NaN"):
            MatchingSolverDualObjectiveFunction(torch_args(q, "f32", pm, DEV), gamma=0.05)
    f = MatchingSolverDualObjectiveFunction(torch_args(p, "f32", pm, DEV), gamma=0.05)
    lam = np.full(p["m"], 0.02)
    f.c.values().mul_(1e7)
    f.costs_changed()
    res = f.calculate(torch.from_numpy(lam).float().to(DEV), gamma=0.05, save_primal=True)
    ax, obj0, ssq, x = oracle.matching_calculate(p["m"], p["n"], p["colptr"], p["rowidx"], p["a"], p["c"] * 1e7, lam, 0.05, [("box", {"lower": 0.0, "upper": 1.0})], dtype=np.float32)
    grad, obj, reg, *_ = agd_oracle.epilogue(ax, obj0, ssq, lam, p["b"], 0.05, np.float32)
    assert relerr(res.primal_var.cpu().numpy(), x) < RTOL["f32"]
    assert relerr([float(res.dual_objective)], [obj]) < RTOL["f32"] * 10  # (garbage before: |c.x| * 2^shift left the 2^51 window of the conversion)
    f.c.values()[7] = float("nan")
    with pytest.raises((HipLibraryError, RuntimeError, ValueError), match="inf or NaN"):
        f.costs_changed()


@pytest.mark.parametrize("order", ["box_then_simplex", "simplex_then_box", "all_simplex"])
def test_release_inputs_makes_the_handle_self_contained(order):
    """``objective.release_inputs()`` (dl_matching_own_inputs): the handle copies the prefix of a / c / rows its tiles read in place and
    stops borrowing; the caller's tensors can then be freed.  Results are bit-identical (calculate, primal, and a device-resident
    solve), the kept prefix is what the map implies (half the arrays for box-then-simplex, next to nothing for all-simplex, all of it
    when the un-sliced block comes last), value refreshes are refused afterwards."""
    import os

    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent
    from dualip_amd.projections import create_projection_map

    if os.environ.get("DUALIP_HIP_SELL") == "0":
        pytest.skip("what the handle keeps is stated for the 256-wide layout with slices")
    p = _random_problem(500, 60000, 9, seed=21, long_cols=[(7, 300), (59990, 400)])
    n, half = p["n"], p["n"] // 2
    box, simplex = ("box", {"lower": 0.0, "upper": 1.0}), ("simplex", {"z": 1.0})
    first, second = {"box_then_simplex": (box, simplex), "simplex_then_box": (simplex, box), "all_simplex": (simplex, simplex)}[order]
    if order == "all_simplex":  # (one entry: two maps of the same operator would share their key)
        pm = create_projection_map("simplex", {"z": 1.0}, None, indices=range(0, n))
    else:
        pm = {**create_projection_map(first[0], dict(first[1]), None, indices=range(0, half)), **create_projection_map(second[0], dict(second[1]), None, indices=range(half, n))}
    lam = torch.from_numpy(np.random.default_rng(3).uniform(0, 0.05, p["m"])).float().to(DEV)
    kw = dict(max_iter=40, gamma=0.02, initial_step_size=1e-3, max_step_size=0.1, iteration_callback=False)

    def run(release):
        args = torch_args(p, "f32", pm, DEV)
        f = MatchingSolverDualObjectiveFunction(args, gamma=0.02)
        assert f.info()["layout"] == 4 and f.info()["slices"] > 0
        out = None
        if release:
            before = torch.cuda.memory_allocated()
            out = f.release_inputs()
            assert f.A is None and f.c is None
            del args  # the caller drops its tensors too: 8 bytes of values + 8 of int64 row indices per non-zero go away
            torch.cuda.synchronize()
            out["freed"] = before - torch.cuda.memory_allocated()
        r = f.calculate(lam, gamma=0.02, save_primal=True)
        res = AcceleratedGradientDescent(**kw).maximize(f, torch.zeros(p["m"], dtype=torch.float32, device=DEV))
        if release:
            with pytest.raises((RuntimeError, ValueError), match="owns its inputs"):
                f.costs_changed()
            # own-then-fairness (the order dl_matching_own_inputs' own refusal does not cover): the fairness stream would be read at pool
            # offsets by the straggler tiles while f stays in the caller's order -- refused, not silently wrong
            from dualip_amd import _hip

            fv = torch.zeros(f.nnz + 8, dtype=torch.float32, device=DEV)
            with torch.cuda.device(f.device):
                rc = f._lib.dl_matching_set_fairness(f._handle, _hip.ptr(fv), _hip.stream_ptr(f.device))
            assert rc == 4 and "owns its inputs" in _hip.last_error(), (rc, _hip.last_error())  # DL_E_STATE
        return r.dual_gradient.clone(), r.primal_var.clone(), float(r.dual_objective), list(res.dual_objective_log), res.dual_val.clone(), out

    g0, x0, o0, log0, d0, _ = run(False)
    g1, x1, o1, log1, d1, info = run(True)
    assert torch.equal(g0, g1) and torch.equal(x0, x1) and o0 == o1 and log0 == log1 and torch.equal(d0, d1)
    nnz = int(p["colptr"][-1])
    k_half = int(p["colptr"][half])
    if order == "box_then_simplex":  # the long simplex column near the end (a K-lane slice, or a straggler moved to the pool) does not extend the prefix
        assert k_half <= info["kept_elements"] <= k_half + 1024, (info, k_half)
    elif order == "all_simplex":
        assert info["kept_elements"] <= 1280, info  # 256 slots for the padding descriptors' prefetch + at most the two long columns
    else:
        assert info["kept_elements"] == nnz, info
    assert info["freed"] > 0 or order == "simplex_then_box"
