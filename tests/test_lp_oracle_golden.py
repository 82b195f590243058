"""Pins oracle/lp_oracle.py (numpy restatement of the reference's generic-LP objective) against fixture G6, produced by the
reference itself (tests/golden/make_golden_lp.py), and against the reference's own known-answer tests for the PDLP
convergence bound (tests/objectives/test_miplib_objective.py:9-99)."""
import numpy as np

from oracle import agd_oracle, lp_oracle
from tests.helpers import NP_DT, load, lp_small_entries, relerr

TOL = {"f32": 3e-5, "f64": 1e-11}


def _dense(z):
    A = np.zeros((int(z["m"]), int(z["n"])))
    A[z["coo_row"], z["coo_col"]] = z["coo_val"]
    return A


def test_small_lp_calculate_matches_reference():
    z = load("g6_lp_small.npz")
    n = int(z["n"])
    for dn, dt in NP_DT.items():
        lo, hi = lp_oracle.bounds_from_map(n, lp_small_entries(z), dt)
        for form in ("dense", "coo"):
            for ln in ("zero", "rand", "signed"):
                lam = np.zeros(int(z["m"])) if ln == "zero" else z[f"lam_{ln}"]
                for g in (0.01, 0.5):
                    key = f"calc|{form}|{ln}|{g}|{dn}"
                    grad, x, obj, reg, primal = lp_oracle.lp_calculate(z["A"], z["c"], z["b"], lo, hi, lam, g, dt)
                    assert relerr(grad, z[f"{key}|grad"]) < TOL[dn], key
                    assert relerr(x, z[f"{key}|x"]) < TOL[dn], key
                    assert relerr([obj, reg, primal], z[f"{key}|scal"]) < TOL[dn] * 10, key
        rn = z[f"row_norms|{dn}"]
        key = f"calc|jacobi|rand|0.01|{dn}"
        grad, x, obj, reg, primal = lp_oracle.lp_calculate(z["A"], z["c"], z["b"], lo, hi, z["lam_rand"], 0.01, dt, row_norms=rn)
        assert relerr(grad, z[f"{key}|grad"]) < TOL[dn] and relerr(x, z[f"{key}|x"]) < TOL[dn]
        assert relerr([obj, reg, primal], z[f"{key}|scal"]) < TOL[dn] * 10


def test_small_lp_traces_match_reference():
    z = load("g6_lp_small.npz")
    n, m = int(z["n"]), int(z["m"])
    for dn, dt in NP_DT.items():
        lo, hi = lp_oracle.bounds_from_map(n, lp_small_entries(z), dt)
        for name, rn in (("plain", None), ("jacobi", z[f"row_norms|{dn}"])):
            def calc(lam, gamma):
                grad, x, obj, reg, primal = lp_oracle.lp_calculate(z["A"], z["c"], z["b"], lo, hi, lam, gamma, dt, row_norms=rn)
                return grad, obj, x

            r = agd_oracle.maximize(calc, np.zeros(m), 300, 1e-2, initial_step_size=1e-3, max_step_size=0.1, eq_mask=z["eq"], dtype=dt)
            want = z[f"trace|{name}|{dn}|obj_log"]
            head = 40 if dn == "f32" else 120
            assert relerr(r["dual_obj_log"][:head], want[:head]) < (2e-4 if dn == "f32" else 1e-8), (name, dn)
            assert relerr(r["dual_obj_log"], want) < (5e-2 if dn == "f32" else 1e-3), (name, dn)
            assert relerr(r["step_log"][:head], z[f"trace|{name}|{dn}|step_log"][:head]) < (1e-3 if dn == "f32" else 1e-7)


def test_convergence_bound_matches_reference():
    z = load("g6_lp_small.npz")
    n = int(z["n"])
    lower, upper = np.full(n, np.nan), np.full(n, np.nan)
    for kind, params, idx in lp_small_entries(z):
        if kind == "box" and not params:
            params = {"lower": 0.0, "upper": 1.0}
        lower[idx] = params.get("lower", np.nan)
        upper[idx] = params.get("upper", np.nan)
    for dn, dt in NP_DT.items():
        for ln in ("rand", "signed"):
            got = lp_oracle.convergence_bound(z["A"], z["c"], z["b"], lower, upper, z[f"lam_{ln}"], x=z[f"bound_x|{dn}"], optimal_primal_obj=-1.25,
                                              tol=1e-2, eq_mask=z["eq"], dtype=dt)
            want = z[f"bound|{ln}|{dn}"]
            assert np.allclose(got[:4], want[:4], rtol=2e-4 if dn == "f32" else 1e-10), (ln, dn, got, want)
            assert float(got[4]) == want[4]


def test_convergence_bound_known_answers():
    # reference tests/objectives/test_miplib_objective.py:9-58 (general bounds) and :61-99 (unit box)
    A = np.array([[1.0, 1.0, 1.0, 0.0], [2.0, -1.0, 0.0, 1.0], [-1.0, 0.0, 4.0, -1.0]])
    b, c = np.array([5.0, 3.0, 2.0]), np.array([2.0, 3.0, -1.0, 4.0])
    lower, upper = np.array([0.0, 1.0, 0.0, -2.0]), np.array([3.0, 4.0, np.nan, 2.0])
    assert lp_oracle.convergence_bound(A, c, b, lower, upper, [0.0, 0.0, 0.25], tol=1e-5)[4]
    assert lp_oracle.convergence_bound(A, c, b, lower, upper, [0.0, -0.01, 0.26], tol=1e-1)[4]
    assert not lp_oracle.convergence_bound(A, c, b, lower, upper, [0.0, -0.01, 0.26], tol=1e-5)[4]
    A, b, c = np.array([[2.0, 0.0], [0.0, 1.0]]), np.array([1.0, 3.0]), np.array([1.0, 1.0])
    lower, upper = np.zeros(2), np.ones(2)
    assert not lp_oracle.convergence_bound(A, c, b, lower, upper, [0.1, 0.1], tol=1e-5)[4]
    assert lp_oracle.convergence_bound(A, c, b, lower, upper, [0.1, 0.1], tol=1)[4]
    assert lp_oracle.convergence_bound(A, c, b, lower, upper, [0.0, 0.0], tol=1e-8)[4]


def test_miplib_instance_matches_reference():
    z = load("g6_miplib_v150.npz")
    A = _dense(z)
    n, m = int(z["n"]), int(z["m"])
    assert (m, n, len(z["coo_val"])) == (7822, 150, 103991)
    for dn, dt in NP_DT.items():
        lo, hi = z["lower"].astype(dt), z["upper"].astype(dt)
        for ln, lam in (("zero", np.zeros(m)), ("rand", z["lam_rand"])):
            grad, x, obj, reg, primal = lp_oracle.lp_calculate(A, z["c"], z["b"], lo, hi, lam, 1e-3, dt)
            key = f"calc|{ln}|{dn}"
            assert relerr(grad, z[f"{key}|grad"]) < TOL[dn] * 4 and relerr(x, z[f"{key}|x"]) < TOL[dn] * 4, key
            assert relerr([obj, reg, primal], z[f"{key}|scal"]) < TOL[dn] * 40, key

    def calc(lam, gamma):
        grad, x, obj, reg, primal = lp_oracle.lp_calculate(A, z["c"], z["b"], lo64, hi64, lam, gamma, np.float64)
        return grad, obj, x

    lo64, hi64 = z["lower"], z["upper"]
    r = agd_oracle.maximize(calc, np.zeros(m), 150, 1e-3, initial_step_size=1e-5, max_step_size=0.1, dtype=np.float64)
    # this instance amplifies round-off by ~1e6 per 20 iterations once the bounds become active (measured: 1e-12 at
    # iteration 40, 2e-6 at 60, 4e-3 at 80): tight while the two runs are the same computation, loose afterwards
    assert relerr(r["dual_obj_log"][:40], z["trace|f64|obj_log"][:40]) < 1e-10
    assert relerr(r["dual_obj_log"], z["trace|f64|obj_log"][:150]) < 5e-3
    # the driver's loose sanity check (examples/miplib_2017/solve_miplib_dataset.py:74) holds on the stored full traces
    for dn in NP_DT:
        assert abs(27 - z[f"trace|{dn}|obj_log"][-1]) < 1


def test_equality_known_answer_and_one_sided_box_default():
    """The reference's tests/test_equality_constraints.py:18-61 (min x1 + 2 x2, x1 + x2 = 4, box {"upper": 1} on x1, x2 in no entry;
    optimum 7.0): the oracle walks the reference's own 1000-iteration trace (fixture g6_lp_warm.npz) and ends at 7.0 +- 1e-5 * (1 + 7).
    ``{"upper": 1}`` keeps BoxProjection's default lower bound 0 (box.py:12-13)."""
    z = load("g6_lp_warm.npz")
    A, c, b = np.array([[1.0, 1.0]]), np.array([1.0, 2.0]), np.array([4.0])
    lo, hi = lp_oracle.bounds_from_map(2, [("box", {"upper": 1}, [0])], np.float32)
    assert (lo[0], hi[0], lo[1], hi[1]) == (0.0, 1.0, -np.inf, np.inf)

    def calc(lam, gamma):
        grad, x, obj, reg, primal = lp_oracle.lp_calculate(A, c, b, lo, hi, lam, gamma, np.float32)
        return grad, obj, x

    r = agd_oracle.maximize(calc, np.zeros(1), 1000, 1e-5, eq_mask=np.array([True]), dtype=np.float32)
    assert abs(r["dual_obj_log"][-1] - 7.0) < 1e-5 + 1e-5 * 7.0  # torch.isclose(atol=1e-5) of the reference's test (rtol 1e-5 default)
    assert relerr(r["dual_obj_log"], z["eq2|obj_log"]) < 1e-5
    assert relerr(r["dual_val"], z["eq2|lam"]) < 1e-5 and r["dual_val"][0] < 0  # an equality row's dual may be negative (agd.py:13-21)


def test_lp_warm_start_matches_reference():
    """BASELINE config 5 "with warm start": the optimiser restarted from the reference's saved duals (run_solver.py:127-132) walks the
    reference's warm trace; fixture g6_lp_warm.npz, 40 x 60 LP with equality rows."""
    z = load("g6_lp_warm.npz")
    zs = load("g6_lp_small.npz")
    n, m = int(zs["n"]), int(zs["m"])
    n_cold, n_warm, gamma, s0 = z["small|params"]
    for dn, dt in NP_DT.items():
        lo, hi = lp_oracle.bounds_from_map(n, lp_small_entries(zs), dt)

        def calc(lam, g):
            grad, x, obj, reg, primal = lp_oracle.lp_calculate(zs["A"], zs["c"], zs["b"], lo, hi, lam, g, dt)
            return grad, obj, x

        r = agd_oracle.maximize(calc, z[f"small|{dn}|cold_lam"], int(n_warm), float(gamma), initial_step_size=float(s0), max_step_size=0.1, eq_mask=zs["eq"], dtype=dt)
        want = z[f"small|{dn}|warm_obj_log"]
        assert relerr(r["dual_obj_log"][:30], want[:30]) < (2e-4 if dn == "f32" else 1e-8), dn
        assert relerr(r["dual_obj_log"], want) < (5e-2 if dn == "f32" else 1e-3), dn
        assert want[0] > z[f"small|{dn}|cold_obj_log"][0]  # a warm start does not begin at the cold start's objective
