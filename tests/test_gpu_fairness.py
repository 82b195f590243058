"""Matching LP with two fairness rows (docs/demo/matching_complex.rst of the reference) on the GPU: parity with the
fixture gf_fairness.npz (reference operators) and with oracle/fairness_oracle.py, one calculate() and whole AGD solves."""
import numpy as np
import pytest
import torch

from tests.helpers import NP_DT, RTOL, SINGLE_MAPS, load, problem, relerr, torch_args

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TD = {"f32": torch.float32, "f64": torch.float64}


def _objective(p, dn, mn, delta, ratio, **kw):
    from dualip_amd.objectives.matching_fairness import MatchingFairnessDualObjectiveFunction
    from dualip_amd.projections import create_projection_map

    kind, params = SINGLE_MAPS[mn]
    args = torch_args(p, dn, create_projection_map(kind, params, p["n"]), DEV)
    args.b_vec = torch.cat([args.b_vec, torch.tensor([delta, delta], dtype=TD[dn], device=DEV)])
    return MatchingFairnessDualObjectiveFunction(args, gamma=0.02, group_ratio=ratio, **kw)


@pytest.mark.parametrize("native", [True, False])
@pytest.mark.parametrize("dn", ["f32", "f64"])
@pytest.mark.parametrize("mn", ["simplex1", "box01"])
def test_calculate_matches_reference_operators(dn, mn, native):
    z = load("gf_fairness.npz")
    p = problem(load("g1_syn2000.npz"))
    f = _objective(p, dn, mn, float(z["delta"]), float(z["group_ratio"]), native=native)
    assert f.native == native
    assert relerr(f._f.cpu().numpy(), z[f"f|{dn}"]) < (1e-7 if dn == "f32" else 1e-15)
    for ln in ("zero", "rand", "tilt"):
        lam = torch.from_numpy(z[f"lam_{ln}"]).to(TD[dn]).to(DEV)
        r = f.calculate(lam, 0.02, save_primal=True)
        pre = f"calc|{mn}|{ln}|{dn}"
        assert relerr(r.primal_var.cpu().numpy(), z[pre + "|x"]) < RTOL[dn], pre
        assert relerr(r.dual_gradient.cpu().numpy(), z[pre + "|grad"]) < RTOL[dn], pre
        assert relerr([float(r.dual_objective), float(r.reg_penalty), float(r.primal_objective)], z[pre + "|scal"]) < RTOL[dn], pre
        g = r.dual_gradient.cpu().numpy().astype(np.float64)
        assert abs((g[-2] + float(z["delta"])) + (g[-1] + float(z["delta"]))) < 1e-6  # the two rows are each other's negation


@pytest.mark.parametrize("native", [True, False])
@pytest.mark.parametrize("mn", ["simplex1", "box01"])
def test_agd_solve_matches_reference_trace(mn, native):
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent

    z = load("gf_fairness.npz")
    p = problem(load("g1_syn2000.npz"))
    f = _objective(p, "f64", mn, float(z["delta"]), float(z["group_ratio"]), native=native)
    solver = AcceleratedGradientDescent(max_iter=60, gamma=0.02, initial_step_size=1e-3, max_step_size=0.1, save_primal=True, iteration_callback=False)
    res = solver.maximize(f, torch.zeros(p["m"] + 2, dtype=torch.float64, device=DEV))
    pre = f"trace|{mn}|f64"
    assert relerr(res.dual_objective_log[:40], z[pre + "|obj_log"][:40]) < 1e-8
    assert np.allclose(res.step_size_log[:40], z[pre + "|step_log"][:40], rtol=1e-5)
    assert relerr(res.dual_objective_log, z[pre + "|obj_log"]) < 2e-2  # chaotic tail (see test_oracle_golden)
    lam = res.dual_val.cpu().numpy()
    assert lam[-2] > 0 and lam[-1] == 0
    assert res.objective_result.primal_var.shape == (len(p["a"]),)
    if mn == "simplex1":
        assert relerr(lam, z[pre + "|lam"]) < 1e-4 and relerr(res.objective_result.primal_var.cpu().numpy(), z[pre + "|x"]) < 1e-3


def test_custom_coefficients_against_the_oracle_and_argument_checks():
    """A_fairness handed in (random signs, not a scaled copy of A), fp32, against the oracle; unbounded maps and a short
    b_vec are refused."""
    from dualip_amd.objectives.matching_fairness import MatchingFairnessDualObjectiveFunction
    from dualip_amd.projections import create_projection_map
    from oracle import fairness_oracle

    p = problem(load("g1_syn2000.npz"))
    rng = np.random.default_rng(8)
    fv = (rng.choice([-1.0, 1.0], size=len(p["a"])) * rng.uniform(0, 2e-3, size=len(p["a"]))).astype(np.float32)
    lam = np.concatenate([rng.uniform(0, 0.01, p["m"]), [0.5, 0.1]]).astype(np.float32)
    b_full = np.concatenate([p["b"], [0.01, 0.01]])
    args = torch_args(p, "f32", create_projection_map("simplex", {"z": 1.0}, p["n"]), DEV)
    short = args.b_vec
    args.b_vec = torch.from_numpy(b_full).float().to(DEV)
    grad, obj, reg, primal, x = fairness_oracle.fairness_calculate(p, fv, lam, 0.02, ("simplex", {"z": 1.0}), b_full, np.float32)
    for native in (True, False):
        f = MatchingFairnessDualObjectiveFunction(args, gamma=0.02, A_fairness=torch.from_numpy(fv).to(DEV), native=native)
        r = f.calculate(torch.from_numpy(lam).to(DEV), save_primal=True)
        assert relerr(r.primal_var.cpu().numpy(), x) < RTOL["f32"]
        assert relerr(r.dual_gradient.cpu().numpy(), grad) < RTOL["f32"]
        assert relerr([float(r.dual_objective), float(r.reg_penalty), float(r.primal_objective)], [obj, reg, primal]) < RTOL["f32"]
    args.b_vec = short
    with pytest.raises(ValueError, match="entries"):
        MatchingFairnessDualObjectiveFunction(args, gamma=0.02)
    cone = torch_args(p, "f32", create_projection_map("cone", {"lower": 0.0}, p["n"]), DEV)
    cone.b_vec = torch.from_numpy(b_full).float().to(DEV)
    with pytest.raises(NotImplementedError, match="bound x"):
        MatchingFairnessDualObjectiveFunction(cone, gamma=0.02, native=False)
    # the kernel form has no such restriction: one-sided bounds against the oracle
    f = MatchingFairnessDualObjectiveFunction(cone, gamma=0.02, A_fairness=torch.from_numpy(fv).to(DEV), native=True)
    r = f.calculate(torch.from_numpy(lam).to(DEV), save_primal=True)
    grad, obj, reg, primal, x = fairness_oracle.fairness_calculate(p, fv, lam, 0.02, ("cone", {"lower": 0.0}), b_full, np.float32)
    assert relerr(r.primal_var.cpu().numpy(), x) < RTOL["f32"] and relerr(r.dual_gradient.cpu().numpy(), grad) < RTOL["f32"]


@pytest.mark.parametrize("hot", [False, True])
def test_kernel_form_through_single_column_tiles_and_the_hot_rows_plan(hot, monkeypatch):
    """The fairness stream through every path of the fused kernel: window tiles, columns walked by one wavefront (300-700
    non-zeros) and by a whole workgroup (3 000), with the dual vector wholly in LDS or under the hot-rows plan (the two dense
    rows are never in the row indices, so they sit in its cold tail); one calculate() and a device-resident AGD solve."""
    from dualip_amd.objectives.matching_fairness import MatchingFairnessDualObjectiveFunction
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent
    from dualip_amd.projections import create_projection_map
    from oracle import agd_oracle, fairness_oracle
    from tests.test_gpu_edge_cases import _random_problem

    if hot:
        monkeypatch.setenv("DUALIP_HIP_HOT_ROWS", "512")
    monkeypatch.setenv("DUALIP_HIP_FLAT", "0")  # the walkers are the subject: keep long point-wise columns out of the window stream
    p = _random_problem(3_500, 2_500, 10, seed=91, long_cols=((3, 400), (1200, 700), (2400, 3000), (2499, 260)), empty_every=23)
    m, n = p["m"], p["n"]
    rng = np.random.default_rng(12)
    b_full = np.concatenate([p["b"], [0.0, 0.0]])
    lam = np.concatenate([rng.uniform(0, 0.01, m), [0.4, 0.1]])
    for dn in ("f32", "f64"):
        td = TD[dn]
        fv = (rng.choice([-1.0, 1.0], size=len(p["a"])) * rng.uniform(0, 5e-3, size=len(p["a"]))).astype(NP_DT[dn])
        for pt, pp in (("simplex", {"z": 1.0}), ("box", {"lower": 0.0, "upper": 0.5})):
            args = torch_args(p, dn, create_projection_map(pt, dict(pp), n), DEV)
            args.b_vec = torch.from_numpy(b_full).to(td).to(DEV)
            f = MatchingFairnessDualObjectiveFunction(args, gamma=0.05, A_fairness=torch.from_numpy(fv).to(DEV), native=True)
            info = f.inner.info()
            assert info["long_columns"] >= 4 and info["workgroup_columns"] == 1 and info["hot_rows"] == (512 if hot else 0)
            r = f.calculate(torch.from_numpy(lam).to(td).to(DEV), save_primal=True)
            grad, obj, reg, primal, x = fairness_oracle.fairness_calculate(p, fv, lam, 0.05, (pt, pp), b_full, NP_DT[dn])
            assert relerr(r.primal_var.cpu().numpy(), x) < RTOL[dn], (dn, pt)
            assert relerr(r.dual_gradient.cpu().numpy(), grad) < RTOL[dn], (dn, pt)
            assert relerr([float(r.dual_objective), float(r.reg_penalty), float(r.primal_objective)], [obj, reg, primal]) < RTOL[dn] * 10, (dn, pt)
    # device-resident solve (fp64) against the oracle's maximiser
    fv = (rng.choice([-1.0, 1.0], size=len(p["a"])) * rng.uniform(0, 5e-3, size=len(p["a"])))
    args = torch_args(p, "f64", create_projection_map("simplex", {"z": 1.0}, n), DEV)
    args.b_vec = torch.from_numpy(b_full).to(DEV)
    f = MatchingFairnessDualObjectiveFunction(args, gamma=0.05, A_fairness=torch.from_numpy(fv).to(DEV), native=True)
    res = AcceleratedGradientDescent(max_iter=40, gamma=0.05, initial_step_size=1e-4, max_step_size=1e-2, iteration_callback=False).maximize(
        f, torch.zeros(m + 2, dtype=torch.float64, device=DEV))

    def calc(lam_, gamma):
        grad, obj, _, _, _ = fairness_oracle.fairness_calculate(p, fv, lam_, gamma, ("simplex", {"z": 1.0}), b_full, np.float64)
        return grad, obj, None

    want = agd_oracle.maximize(calc, np.zeros(m + 2), 40, 0.05, initial_step_size=1e-4, max_step_size=1e-2, dtype=np.float64)
    assert relerr(res.dual_objective_log, want["dual_obj_log"]) < 1e-8
    assert relerr(res.dual_val.cpu().numpy(), want["dual_val"]) < 1e-8
