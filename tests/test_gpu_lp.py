"""GPU parity tests of the generic-LP ("miplib2017") objective (run with ``-m gpu`` on an MI355X): the HIP kernels behind
``dl_lp_*`` against fixture G6 (produced by the reference itself, tests/golden/make_golden_lp.py) and the numpy oracle.

Tolerances: one ``calculate`` -- 3e-5 (fp32) / 1e-11 (fp64) relative to the largest magnitude of the compared vector
(summation order is the only difference); AGD traces -- tight over the first iterations, loose once the instance has
amplified round-off (see tests/test_lp_oracle_golden.py for the measured growth).
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import lp_oracle
from tests.helpers import NP_DT, load, lp_small_entries, relerr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TD = {"f32": torch.float32, "f64": torch.float64}
TOL = {"f32": 3e-5, "f64": 1e-11}


def _small_map(z, names=("lower", "upper")):
    from dualip_amd.projections.base import ProjectionEntry

    lo, up = names
    out = {}
    for k, (kind, params, idx) in zip(("two", "unit", "lo", "up", "lu"), lp_small_entries(z)):
        if kind == "box" and params:
            params = {lo: params["lower"], up: params["upper"]}
        out[k] = ProjectionEntry(kind, dict(params), indices=[int(i) for i in idx])
    return out


def _small_objective(z, dn, form="coo", jacobi=False, pm=None, eq=True):
    from dualip_amd.objectives.miplib import MIPLIB2017ObjectiveFunction, MIPLIBInputArgs

    dt = TD[dn]
    A = torch.from_numpy(z["A"]).to(dt)
    A = {"dense": A, "coo": A.to_sparse_coo(), "csr": A.to_sparse_csr(), "csc": A.to_sparse_csc()}[form].to(DEV)
    args = MIPLIBInputArgs(
        A=A, c=torch.from_numpy(z["c"]).to(dt).to(DEV), b_vec=torch.from_numpy(z["b"]).to(dt).to(DEV),
        projection_map=pm if pm is not None else _small_map(z), equality_mask=torch.from_numpy(z["eq"]).to(DEV) if eq else None,
    )
    return MIPLIB2017ObjectiveFunction(miplib_input_args=args, use_jacobi_precondition=jacobi)


def _check_calc(res, z, key, dn, scale=1.0):
    assert relerr(res.dual_gradient.cpu().numpy(), z[f"{key}|grad"]) < TOL[dn] * scale, key
    assert relerr(res.primal_var.cpu().numpy(), z[f"{key}|x"]) < TOL[dn] * scale, key
    got = [float(res.dual_objective), float(res.reg_penalty), float(res.primal_objective)]
    assert relerr(got, z[f"{key}|scal"]) < TOL[dn] * 10 * scale, (key, got, z[f"{key}|scal"])


@pytest.mark.parametrize("form", ["dense", "coo", "csr", "csc"])
def test_small_lp_calculate_matches_reference_golden(form):
    z = load("g6_lp_small.npz")
    for dn in NP_DT:
        f = _small_objective(z, dn, form)
        gform = form if form in ("dense", "coo") else "coo"
        for ln in ("zero", "rand", "signed"):
            lam = np.zeros(int(z["m"])) if ln == "zero" else z[f"lam_{ln}"]
            for g in (0.01, 0.5):
                res = f.calculate(torch.from_numpy(lam).to(TD[dn]).to(DEV), g, save_primal=True)
                _check_calc(res, z, f"calc|{gform}|{ln}|{g}|{dn}", dn)


def test_l_u_spelling_and_jacobi():
    z = load("g6_lp_small.npz")
    for dn in NP_DT:
        f = _small_objective(z, dn, pm=_small_map(z, names=("l", "u")))
        res = f.calculate(torch.from_numpy(z["lam_rand"]).to(TD[dn]).to(DEV), 0.01, save_primal=True)
        _check_calc(res, z, f"calc|coo|rand|0.01|{dn}", dn)
        for form in ("dense", "coo"):  # (the reference only preconditions dense A; sparse gives the same numbers here)
            fj = _small_objective(z, dn, form, jacobi=True)
            assert relerr(fj.row_norms.cpu().numpy(), z[f"row_norms|{dn}"]) < TOL[dn]
            res = fj.calculate(torch.from_numpy(z["lam_rand"]).to(TD[dn]).to(DEV), 0.01, save_primal=True)
            _check_calc(res, z, f"calc|jacobi|rand|0.01|{dn}", dn, scale=4.0)
            lam, grad = fj.invert_jacobi_precondition(torch.ones(int(z["m"]), device=DEV, dtype=TD[dn]), res.dual_gradient)
            assert relerr(lam.cpu().numpy(), 1 / z[f"row_norms|{dn}"]) < TOL[dn]


@pytest.mark.parametrize("name", ["plain", "jacobi"])
def test_small_lp_traces_match_reference_golden(name):
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent

    z = load("g6_lp_small.npz")
    m = int(z["m"])
    for dn in NP_DT:
        f = _small_objective(z, dn, "dense" if name == "jacobi" else "coo", jacobi=(name == "jacobi"))
        solver = AcceleratedGradientDescent(max_iter=300, gamma=1e-2, initial_step_size=1e-3, max_step_size=0.1, save_primal=True, iteration_callback=False)
        res = solver.maximize(f, torch.zeros(m, dtype=TD[dn], device=DEV))
        want = z[f"trace|{name}|{dn}|obj_log"]
        head = 40 if dn == "f32" else 120
        assert relerr(res.dual_objective_log[:head], want[:head]) < (2e-4 if dn == "f32" else 1e-8), (name, dn)
        assert relerr(res.dual_objective_log, want) < (5e-2 if dn == "f32" else 1e-3), (name, dn)
        assert relerr(res.step_size_log[:head], z[f"trace|{name}|{dn}|step_log"][:head]) < (1e-3 if dn == "f32" else 1e-7)
        assert res.objective_result.primal_var.shape == (int(z["n"]),)
        if dn == "f64":
            assert relerr(res.dual_val.cpu().numpy(), z[f"trace|{name}|{dn}|lam"]) < 1e-2
            assert relerr(res.objective_result.primal_var.cpu().numpy(), z[f"trace|{name}|{dn}|x"]) < 5e-2  # iterate 300: amplified round-off


def test_generic_route_agrees_with_device_route():
    """The torch route of the maximizer (any BaseObjective) driving the same objective gives the same trace."""
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent

    z = load("g6_lp_small.npz")
    f = _small_objective(z, "f64")
    kw = dict(max_iter=60, gamma=1e-2, initial_step_size=1e-3, max_step_size=0.1, iteration_callback=False)
    a = AcceleratedGradientDescent(**kw).maximize(f, torch.zeros(int(z["m"]), dtype=torch.float64, device=DEV))
    b = AcceleratedGradientDescent(**kw)._maximize_generic(f, torch.zeros(int(z["m"]), dtype=torch.float64, device=DEV), 0)
    assert relerr(a.dual_objective_log, b.dual_objective_log) < 1e-9
    assert relerr(a.dual_val.cpu().numpy(), b.dual_val.cpu().numpy()) < 1e-9


def test_operator_on_an_index_set():
    """A registered operator that is not a point-wise bound (simplex over variables 0..19): two-call route.  The reference
    fails on this map (miplib.py:90, shape mismatch), so the expected values come from the oracle's pieces."""
    from dualip_amd.projections.base import ProjectionEntry

    z = load("g6_lp_small.npz")
    pm = {
        "s": ProjectionEntry("simplex", {"z": 2.0}, indices=list(range(0, 20))),
        "b": ProjectionEntry("box", {"lower": 0.0, "upper": 1.0}, indices=list(range(20, 40))),
    }
    n, m = int(z["n"]), int(z["m"])
    for dn, dt in NP_DT.items():
        f = _small_objective(z, dn, pm=pm, eq=False)
        res = f.calculate(torch.from_numpy(z["lam_rand"]).to(TD[dn]).to(DEV), 0.01, save_primal=True)
        free = lp_oracle.lp_calculate(z["A"], z["c"], z["b"], np.full(n, -np.inf), np.full(n, np.inf), z["lam_rand"], 0.01, dt)[1]
        x = free.copy()
        x[:20] = oracle.project_dense(free[:20].reshape(-1, 1).astype(dt), "simplex", {"z": 2.0}).reshape(-1)
        x[20:40] = np.clip(free[20:40], 0.0, 1.0)
        grad = (np.asarray(z["A"], dtype=dt) @ x - z["b"].astype(dt)).astype(dt)
        assert relerr(res.primal_var.cpu().numpy(), x) < TOL[dn] * 4
        assert relerr(res.dual_gradient.cpu().numpy(), grad) < TOL[dn] * 4
        assert abs(float(res.primal_var[:20].sum()) - 2.0) < 1e-5


def test_convergence_bound_matches_reference_golden():
    from dualip_amd.projections.base import ProjectionEntry

    z = load("g6_lp_small.npz")
    for dn in NP_DT:
        f = _small_objective(z, dn, "dense", pm=_small_map(z, names=("l", "u")))
        for ln in ("rand", "signed"):
            got = f.calculate_convergence_bound(torch.from_numpy(z[f"lam_{ln}"]).to(TD[dn]).to(DEV), x=torch.from_numpy(z[f"bound_x|{dn}"]).to(DEV),
                                                optimal_primal_obj=-1.25, tol=1e-2)
            want = z[f"bound|{ln}|{dn}"]
            assert np.allclose([float(v) for v in got[:4]], want[:4], rtol=3e-4 if dn == "f32" else 1e-9), (ln, dn, got, want)
            assert float(bool(got[4])) == want[4]
    # known answers of the reference's tests (tests/objectives/test_miplib_objective.py:9-58)
    from dualip_amd.objectives.miplib import MIPLIB2017ObjectiveFunction, MIPLIBInputArgs

    A = torch.tensor([[1.0, 1.0, 1.0, 0.0], [2.0, -1.0, 0.0, 1.0], [-1.0, 0.0, 4.0, -1.0]], device=DEV)
    pm = {
        "bound_1": ProjectionEntry("box", {"l": 0.0, "u": 3.0}, indices=[0]),
        "bound_2": ProjectionEntry("box", {"l": 1.0, "u": 4.0}, indices=[1]),
        "bound_3": ProjectionEntry("box", {"l": 0.0, "u": float("nan")}, indices=[2]),
        "bound_4": ProjectionEntry("box", {"l": -2.0, "u": 2.0}, indices=[3]),
    }
    args = MIPLIBInputArgs(A=A, c=torch.tensor([2.0, 3.0, -1.0, 4.0], device=DEV), projection_map=pm, b_vec=torch.tensor([5.0, 3.0, 2.0], device=DEV),
                           equality_mask=torch.tensor([False, False, False], device=DEV))
    f = MIPLIB2017ObjectiveFunction(miplib_input_args=args)
    assert f.calculate_convergence_bound(torch.tensor([0.0, 0.0, 0.25], device=DEV), tol=1e-5)[4]
    assert f.calculate_convergence_bound(torch.tensor([0.0, -0.01, 0.26], device=DEV), tol=1e-1)[4]
    assert not f.calculate_convergence_bound(torch.tensor([0.0, -0.01, 0.26], device=DEV), tol=1e-5)[4]


def _miplib_args(z, dn):
    from dualip_amd.objectives.miplib import MIPLIBInputArgs
    from dualip_amd.projections.base import ProjectionEntry

    dt = TD[dn]
    m, n = int(z["m"]), int(z["n"])
    idx = torch.from_numpy(np.stack([z["coo_row"], z["coo_col"]]).astype(np.int64))
    A = torch.sparse_coo_tensor(idx, torch.from_numpy(z["coo_val"]).to(dt), (m, n))
    groups = {}
    for j, bd in enumerate(zip(z["lower"], z["upper"])):
        groups.setdefault(bd, []).append(j)
    pm = {f"bound_{bd}": ProjectionEntry("box", {"lower": float(bd[0]), "upper": float(bd[1])}, indices=ix) for bd, ix in groups.items()}  # read_mps_data.py:173-188
    eq = torch.from_numpy(z["equality_mask"]) if z["equality_mask"].any() else None
    return MIPLIBInputArgs(A=A, c=torch.from_numpy(z["c"]).to(dt), b_vec=torch.from_numpy(z["b"]).to(dt), projection_map=pm, equality_mask=eq)


def test_miplib_instance_calculate_and_trace():
    """BASELINE config 5: examples/miplib_2017/v150d30-2hopcds through run_solver (solve_miplib_dataset.py:45-75)."""
    from dualip_amd.objectives.miplib import MIPLIB2017ObjectiveFunction
    from dualip_amd.run_solver import run_solver, transfer_tensors_to_device
    from dualip_amd.types import ComputeArgs, ObjectiveArgs, SolverArgs

    z = load("g6_miplib_v150.npz")
    m = int(z["m"])
    for dn in NP_DT:
        args = _miplib_args(z, dn)
        f = MIPLIB2017ObjectiveFunction(miplib_input_args=transfer_tensors_to_device(args, DEV))
        for ln, lam in (("zero", np.zeros(m)), ("rand", z["lam_rand"])):
            res = f.calculate(torch.from_numpy(lam).to(TD[dn]).to(DEV), 1e-3, save_primal=True)
            _check_calc(res, z, f"calc|{ln}|{dn}", dn, scale=4.0)
        import contextlib
        import io

        with contextlib.redirect_stdout(io.StringIO()):
            out = run_solver(args, SolverArgs(max_iter=2000, initial_step_size=1e-5, gamma=1e-3), ComputeArgs(host_device=DEV), ObjectiveArgs(objective_type="miplib2017"))
        want = z[f"trace|{dn}|obj_log"]
        log = np.array(out.dual_objective_log)
        assert len(log) == 2000
        head = 25 if dn == "f32" else 40
        assert relerr(log[:head], want[:head]) < (2e-5 if dn == "f32" else 1e-9), dn
        assert relerr(log, want) < 2e-2, (dn, np.abs(log - want).max())
        assert abs(27 - out.dual_objective) < 1  # the driver's own sanity check (solve_miplib_dataset.py:74)
        assert abs(log[99] - want[99]) < 0.05 and abs(log[999] - want[999]) < 0.2 and abs(log[1999] - want[1999]) < 0.3


def test_miplib_instance_from_the_mps_file():
    """BASELINE config 5 from the shipped ``.mps.gz`` itself (solve_miplib_dataset.py:19-75: read_mps_file -> to_dualip_format ->
    MIPLIBInputArgs -> run_solver): the reader of this package, then the same trace as the reference's (fixture G6)."""
    import contextlib
    import io
    import os

    from dualip_amd.objectives.miplib import MIPLIBInputArgs
    from dualip_amd.run_solver import run_solver
    from dualip_amd.types import ComputeArgs, ObjectiveArgs, SolverArgs
    from dualip_amd.utils.read_mps_data import read_mps_file

    z = load("g6_miplib_v150.npz")
    data = read_mps_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "v150d30-2hopcds.mps.gz")).to_dualip_format()  # float32, as the driver
    args = MIPLIBInputArgs(A=data.A, c=data.C, b_vec=data.b_vec, projection_map=data.projection_map, equality_mask=data.equality_mask)
    with contextlib.redirect_stdout(io.StringIO()):
        out = run_solver(args, SolverArgs(max_iter=2000, initial_step_size=1e-5, gamma=1e-3), ComputeArgs(host_device=DEV), ObjectiveArgs(objective_type="miplib2017"))
    want = z["trace|f32|obj_log"]
    log = np.array(out.dual_objective_log)
    assert relerr(log[:25], want[:25]) < 2e-5 and relerr(log, want) < 2e-2
    assert abs(27 - out.dual_objective) < 1  # the driver's own check
    assert abs(log[99] - 23.13099) < 0.05 and abs(log[999] - 25.60996) < 0.2 and abs(log[1999] - 27.01548) < 0.3


def test_equality_constraint_known_answer():
    """The reference's tests/test_equality_constraints.py re-expressed (same numbers, own code), on the device:
    :8-15 the masked projection on the non-negative cone, exact; :18-61 min x1 + 2 x2 s.t. x1 + x2 = 4, 0 <= x1 <= 1 (a box entry that
    only names ``upper``: BoxProjection's default lower bound 0 stays, box.py:12-13), x2 in no entry -- MIPLIB2017ObjectiveFunction +
    AcceleratedGradientDescent(max_iter=1000, gamma=1e-5) reach 7.0 within torch.isclose(atol=1e-5), and walk the reference's own
    trace (fixture g6_lp_warm.npz: the equality row's dual goes to -2, which only the masked projection allows)."""
    from dualip_amd.objectives.miplib import MIPLIB2017ObjectiveFunction, MIPLIBInputArgs
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent, project_on_nn_cone
    from dualip_amd.projections.base import create_projection_map

    y = torch.tensor([-1.0, -1.0, 2.0, -3.0, 4.0], device=DEV)
    mask = torch.tensor([False, True, False, True, False], device=DEV)
    assert (project_on_nn_cone(y, mask) == torch.tensor([0.0, -1.0, 2.0, -3.0, 4.0], device=DEV)).all()

    z = load("g6_lp_warm.npz")
    for A_form in ("dense", "coo"):
        A = torch.tensor([[1.0, 1.0]], device=DEV)
        args = MIPLIBInputArgs(
            A=A if A_form == "dense" else A.to_sparse_coo(), c=torch.tensor([1.0, 2.0], device=DEV), b_vec=torch.tensor([4.0], device=DEV),
            projection_map=create_projection_map("box", {"upper": 1}, num_indices=2, indices=[0]), equality_mask=torch.tensor([True], device=DEV),
        )
        objective = MIPLIB2017ObjectiveFunction(miplib_input_args=args)
        res = AcceleratedGradientDescent(max_iter=1000, gamma=1e-5, iteration_callback=False).maximize(objective, torch.tensor([0.0], device=DEV))
        assert torch.isclose(torch.tensor(res.dual_objective), torch.tensor(7.0), atol=1e-5), res.dual_objective
        assert relerr(res.dual_objective_log, z["eq2|obj_log"]) < 1e-5
        assert relerr(res.step_size_log, z["eq2|step_log"]) < 1e-4
        assert abs(float(res.dual_val[0]) - float(z["eq2|lam"][0])) < 1e-4 and float(res.dual_val[0]) < 0
    # one calculate() at the optimal dual: x = (1, 3) up to the regularisation (x2 free: -(lambda + c2) / gamma)
    r = objective.calculate(torch.tensor([-2.0 - 3e-5], device=DEV), 1e-5, save_primal=True)
    assert float(r.primal_var[0]) == 1.0 and abs(float(r.primal_var[1]) - 3.0) < 0.2
    r0 = objective.calculate(torch.tensor([5.0], device=DEV), 1e-5, save_primal=True)
    assert float(r0.primal_var[0]) == 0.0  # the default lower bound of {"upper": 1}


def _warm_args(z, zs, dn, which):
    if which == "small":
        from dualip_amd.objectives.miplib import MIPLIBInputArgs

        dt = TD[dn]
        return MIPLIBInputArgs(A=torch.from_numpy(zs["A"]).to(dt).to_sparse_coo(), c=torch.from_numpy(zs["c"]).to(dt), b_vec=torch.from_numpy(zs["b"]).to(dt),
                               projection_map=_small_map(zs), equality_mask=torch.from_numpy(zs["eq"]))
    return _miplib_args(zs, dn)


@pytest.mark.parametrize("which", ["small", "v150"])
def test_run_solver_generic_lp_warm_start_matches_reference_golden(which, tmp_path):
    """BASELINE config 5 as worded -- the generic-LP objective WITH a warm start: ``run_solver(objective_type="miplib2017",
    SolverArgs(initial_dual_path=...))`` (run_solver.py:127-132) from the duals the REFERENCE saved after its cold run walks the
    reference's warm trace (fixture g6_lp_warm.npz: its own run_solver, cold then warm); and the chain cold -> save -> warm of this
    package ends where the reference's chain ends."""
    import contextlib
    import io

    from dualip_amd.run_solver import run_solver
    from dualip_amd.types import ComputeArgs, ObjectiveArgs, SolverArgs

    z = load("g6_lp_warm.npz")
    zs = load("g6_lp_small.npz" if which == "small" else "g6_miplib_v150.npz")
    n_cold, n_warm, gamma, s0 = z[f"{which}|params"]
    for dn in NP_DT:
        args = _warm_args(z, zs, dn, which)
        path = str(tmp_path / f"ref_dual_{dn}.pt")
        torch.save(torch.from_numpy(z[f"{which}|{dn}|cold_lam"]), path)
        sa = SolverArgs(max_iter=int(n_warm), gamma=float(gamma), initial_step_size=float(s0), max_step_size=0.1, save_primal=True, initial_dual_path=path)
        with contextlib.redirect_stdout(io.StringIO()):
            warm = run_solver(args, sa, ComputeArgs(host_device=DEV), ObjectiveArgs(objective_type="miplib2017"))
        want = z[f"{which}|{dn}|warm_obj_log"]
        head = 25 if dn == "f32" else 40
        # (same inputs, same computation: tight while round-off has not been amplified -- see test_lp_oracle_golden.py for the growth; a warm
        #  start of the MIPLIB instance begins where its bounds are active, so fp32 runs part at the 1e-5 level from the third iteration on:
        #  measured 5.7e-5 over the first 25, against 2e-4 = this suite's fp32 tolerance for one calculate)
        assert relerr(warm.dual_objective_log[:head], want[:head]) < (2e-4 if dn == "f32" else 1e-8), (which, dn)
        assert relerr(warm.dual_objective_log, want) < (5e-2 if dn == "f32" else 5e-3), (which, dn)
        assert relerr(warm.step_size_log[:head], z[f"{which}|{dn}|warm_step_log"][:head]) < (1e-3 if dn == "f32" else 1e-6)
        assert warm.dual_objective_log[0] > z[f"{which}|{dn}|cold_obj_log"][0]  # it did start from the loaded duals
        # own chain: cold run, torch.save of ITS duals, warm run
        with contextlib.redirect_stdout(io.StringIO()):
            cold = run_solver(args, SolverArgs(max_iter=int(n_cold), gamma=float(gamma), initial_step_size=float(s0), max_step_size=0.1), ComputeArgs(host_device=DEV),
                              ObjectiveArgs(objective_type="miplib2017"))
            own = str(tmp_path / f"own_dual_{dn}.pt")
            torch.save(cold.dual_val.cpu(), own)
            chained = run_solver(args, SolverArgs(max_iter=int(n_warm), gamma=float(gamma), initial_step_size=float(s0), max_step_size=0.1, initial_dual_path=own),
                                 ComputeArgs(host_device=DEV), ObjectiveArgs(objective_type="miplib2017"))
        assert relerr(cold.dual_objective_log, z[f"{which}|{dn}|cold_obj_log"]) < (5e-2 if dn == "f32" else 5e-3)
        assert abs(chained.dual_objective_log[-1] - want[-1]) < 2e-2 * abs(want[-1]), (which, dn)
