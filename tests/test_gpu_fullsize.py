"""BASELINE.json's configurations at 10M entities inside the driver-run suite (``-m gpu``): config 3 (all-simplex map, gamma
continuation 35 / 0.7) and the mixed box / simplex map of config 4, both through the device-resident loop
(``start_device_run``), each followed by the checks bench.py runs at the benchmark size (tests/helpers.verify_at_size: the oracle
on slabs of columns inside every projection block / straddling the block boundary / at the end of the arrays, A x, c.x and
sum x^2 recomputed in float64 from the primal, the sharded route against the single objective).  Plus: two solves of the same
10M problem log IDENTICAL dual objectives while the deal of the window tiles adapts to measured timings (the scalar sums are
integer sums since round 3), and the golden ``calculate`` cases under the alternative tile layout and without slices.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda:0"
N, M = 10_000_000, 10_000


def _problem(kind):
    from benchmark.synthetic import CHUNK_COLS, generate_matching_problem
    from dualip_amd.projections import create_projection_map

    prob = generate_matching_problem(N, M, 1e-3, seed=42, device=DEV, dtype=torch.float32)
    inp = prob["input_args"]
    if kind == "simplex":
        pm = create_projection_map("simplex", {"z": 1.0}, None, indices=range(N))
    else:
        half = (N // 2) // CHUNK_COLS * CHUNK_COLS
        pm = {**create_projection_map("box", {"lower": 0.0, "upper": 1.0}, None, indices=range(0, half)), **create_projection_map("simplex", {"z": 1.0}, None, indices=range(half, N))}
    inp.projection_map = pm
    return inp, pm


def _solve(f, iters, gamma0, decay):
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent

    kw = dict(gamma_decay_type="step", gamma_decay_params={"decay_steps": 35, "decay_factor": 0.7}) if decay else {}
    solver = AcceleratedGradientDescent(max_iter=iters, gamma=gamma0, initial_step_size=1e-3, max_step_size=1e-1, iteration_callback=False, **kw)
    run = solver.start_device_run(f, torch.zeros(M, dtype=torch.float32, device=DEV))
    run.advance(iters)
    res = run.finish()
    run.close()
    return res, float(solver.gamma)


@pytest.mark.parametrize("kind", ["simplex_continuation", "mixed"])
def test_ten_million_entities_through_the_device_loop(kind):
    if os.environ.get("DUALIP_HIP_SELL") == "0":
        pytest.skip("asserts the default kernel plan (256-wide layout with slices)")
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from tests.helpers import verify_at_size

    decay = kind == "simplex_continuation"
    inp, pm = _problem("simplex" if decay else "mixed")
    iters = 200
    gamma0 = 1e-3 / (0.7 ** (iters // 35)) if decay else 1e-3  # run_matching_benchmark.py:29-38: gamma ends at 1e-3
    f = MatchingSolverDualObjectiveFunction(inp, gamma0)
    res, gamma_end = _solve(f, iters, gamma0, decay)
    log = np.array(res.dual_objective_log)
    assert len(log) == iters and np.isfinite(log).all()
    if decay:
        assert abs(gamma_end - 1e-3) < 1e-15 * 10
        assert len(set(np.round(res.step_size_log, 12))) > 2  # the continuation moved the step cap (agd.py:102-109)
    assert log[-1] > log[5]  # dual ascent
    out = verify_at_size("f32", gamma_end, inp, pm, f, f, res.dual_val, device=DEV)
    bad = [c for c in out["checks"] if not c["ok"]]
    assert out["ok"] and not bad, bad
    names = " ".join(c["name"] for c in out["checks"])
    assert "inside entry" in names and "last columns" in names and "recomputed from the primal" in names and "sharded route" in names
    if not decay:
        assert "straddling" in names
    info = f.info()
    assert info["layout"] == 4 and info["slices"] > 0  # the benchmark's kernel plan, not a fallback


def test_hundred_million_entities_headline_under_the_checker():
    """BASELINE config 4's single-GPU size INSIDE the driver-run suite (round-5 review): 100M entities x 10k destinations, mixed box / simplex map,
    generated on the device, 30 iterations of the device loop, then the checker bench.py runs at this size (benchmark/verify.py: oracle slabs of 5000
    columns inside both projection blocks, straddling their boundary and at the END of the arrays -- non-zero offsets up to 10^9 -- A x, c.x and
    sum x^2 recomputed in float64 from the primal, the sharded route against the single objective).  About half a minute and 25 GB of HBM."""
    from benchmark.synthetic import CHUNK_COLS, generate_matching_problem
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.projections import create_projection_map
    from tests.helpers import verify_at_size

    if os.environ.get("DUALIP_HIP_SELL") == "0":
        pytest.skip("asserts the default kernel plan (256-wide layout with slices)")
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2**30:
        pytest.skip("needs 60 GB of free device memory")
    n = 100_000_000
    prob = generate_matching_problem(n, M, 1e-3, seed=42, device=DEV, dtype=torch.float32)
    inp = prob["input_args"]
    half = (n // 2) // CHUNK_COLS * CHUNK_COLS
    pm = {**create_projection_map("box", {"lower": 0.0, "upper": 1.0}, None, indices=range(0, half)), **create_projection_map("simplex", {"z": 1.0}, None, indices=range(half, n))}
    inp.projection_map = pm
    assert prob["nnz"] > 990_000_000
    f = MatchingSolverDualObjectiveFunction(inp, 1e-3)
    info = f.info()
    assert info["layout"] == 4 and info["slices"] > 0 and info["workgroups"] == 256 and info["lambda_in_lds"] == 1 and info["grad_in_lds"] == 1, info
    res, gamma_end = _solve(f, 30, 1e-3, False)
    log = np.array(res.dual_objective_log)
    assert len(log) == 30 and np.isfinite(log).all() and log[-1] > log[5]
    out = verify_at_size("f32", gamma_end, inp, pm, f, f, res.dual_val, device=DEV)
    bad = [c for c in out["checks"] if not c["ok"]]
    assert out["ok"] and not bad, bad
    names = " ".join(c["name"] for c in out["checks"])
    assert "inside entry" in names and "straddling" in names and "last columns" in names and "recomputed from the primal" in names and "sharded route" in names
    del f, inp, prob
    torch.cuda.empty_cache()


def test_logs_are_bit_identical_run_to_run_with_the_adaptive_deal():
    """Two independent handles, two solves each, of the 10M mixed problem: the deal of the window tiles adapts to wall-clock stamps
    (it does at this size: >= 40 rounds per wavefront), yet every logged dual objective, step size and the final duals agree bit
    for bit -- gradient AND scalar sums are integer sums (fused_common.h)."""
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction

    inp, pm = _problem("mixed")
    outs = []
    for _ in range(2):
        f = MatchingSolverDualObjectiveFunction(inp, 1e-3)
        assert f.info()["tiles"] / (16 * f.info()["workgroups"]) >= 40  # the balance is active at this size
        for _ in range(2):
            res, _g = _solve(f, 60, 1e-3, False)
            outs.append((list(res.dual_objective_log), list(res.step_size_log), res.dual_val.clone(), res.objective_result.dual_gradient.clone()))
        del f
    for log, steps, lam, grad in outs[1:]:
        assert log == outs[0][0] and steps == outs[0][1]
        assert torch.equal(lam, outs[0][2]) and torch.equal(grad, outs[0][3])


def test_movielens_shape_at_full_size_under_the_checker():
    """BASELINE config 1 at its REAL shape (examples/movielens_matching/movies_lens_matching.py:222-275: 138 493 users x 26 744
    movies, one simplex per user, capacity 30, gamma 0.1) -- the shape whose kernel plan is everything the benchmark's is not:
    more rows than the LDS holds (hot-rows plan, cold tail on global atomics), the fused kernel's second binary, K-lane slices,
    in-place one-column slices, whole-workgroup columns, all at once in fp32.  50 iterations of the device loop, then the checker
    of bench.py (oracle slabs incl. a column of every length class, sums recomputed in float64, the two-handle route) at the
    solve's duals and at a stress dual vector that multiplies the Newton passes."""
    if os.environ.get("DUALIP_HIP_SELL") == "0":
        pytest.skip("asserts the default kernel plan (256-wide layout with slices)")
    from benchmark.movielens_like import LENGTH_CLASSES, generate, stress_duals
    from dualip_amd.objectives.matching import MatchingInputArgs, MatchingSolverDualObjectiveFunction
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent
    from dualip_amd.projections import create_projection_map
    from tests.helpers import verify_at_size

    A, C, counts = generate(device=DEV)
    n, m = A.shape[1], A.shape[0]
    assert (n, m) == (138_493, 26_744)
    for lo, hi in LENGTH_CLASSES[1:]:
        assert int(((counts >= lo) & (counts <= hi)).sum()) > 0, (lo, hi)  # (generate() draws no user below 20 ratings; after de-duplication a few fall under 25)
    gamma = 0.1
    inp = MatchingInputArgs(A=A, c=C, projection_map=create_projection_map("simplex", {"z": 1.0}, n, indices=range(n)), b_vec=torch.full((m,), 30.0, device=DEV), equality_mask=None)
    f = MatchingSolverDualObjectiveFunction(matching_input_args=inp, gamma=gamma)
    info = f.info()
    assert info["layout"] == 4 and info["hot_rows"] > 0 and info["hot_rows"] < m, info        # more rows than the LDS holds
    assert info["lambda_rows_in_lds"] == m, info  # ... but the whole dual vector fits beside the gradient's hot rows: no tile gathers from L2
    assert info["slice_lane_columns"] > 0 and info["long_columns"] > 0 and info["workgroup_columns"] > 0, info  # second binary: K-lane slices, single-column tiles, whole-workgroup columns
    solver = AcceleratedGradientDescent(max_iter=50, gamma=gamma, initial_step_size=1e-5, max_step_size=1e-3, iteration_callback=False)
    run = solver.start_device_run(f, torch.zeros(m, dtype=torch.float32, device=DEV))
    run.advance(50)
    res = run.finish()
    run.close()
    log = np.array(res.dual_objective_log)
    assert len(log) == 50 and np.isfinite(log).all() and log[-1] > log[0]
    assert float(res.dual_val.abs().max()) > 0
    for tag, lam in (("solve", res.dual_val), ("stress", stress_duals(m, DEV))):
        out = verify_at_size("f32", gamma, inp, inp.projection_map, f, f, lam.contiguous(), device=DEV, length_classes=LENGTH_CLASSES)
        bad = [c for c in out["checks"] if not c["ok"]]
        assert out["ok"] and not bad, (tag, bad)
        names = " ".join(c["name"] for c in out["checks"])
        for lo, hi in LENGTH_CLASSES[1:]:
            assert f"length class [{lo}, {hi}]" in names, (lo, hi)
        assert "recomputed from the primal" in names and "sharded route" in names


@pytest.mark.parametrize("switch", [("DUALIP_HIP_SELL", "0"), ("DUALIP_HIP_LANES_BINARY", "1"), ("DUALIP_HIP_LANES_BINARY", "0"), ("DUALIP_HIP_SELL_LANES", "0"),
                                    ("DUALIP_HIP_FLAT", "0"), ("DUALIP_HIP_FLAT", "1"), ("DUALIP_HIP_COMPACT", "0"), ("DUALIP_HIP_HOST_PACK", "1"), ("DUALIP_HIP_ROW32", "1"),
                                    ("DUALIP_HIP_XCD_BALANCE", "0"), ("DUALIP_HIP_LDS_MODE", "grad"), ("DUALIP_HIP_LDS_MODE", "none"), ("DUALIP_HIP_HOT_ROWS", "64")])
def test_goldens_under_the_alternative_kernel_plans(switch, monkeypatch):
    """The reference's golden ``calculate`` cases (fixture G1) with the column-per-lane slices switched off (every simplex column in
    window tiles); with the fused kernel's second binary (K-lane slices, in-place single-column slices, dynamic deal inside a workgroup)
    forced on and off for the handles free to use either; with one lane per column only; and under every other layout switch of
    INTEGRATION.md (whole-column / unaligned point-wise windows, 12-dword descriptors, host packing, 32-bit rows, the even deal, the
    smaller LDS plans, a forced hot-rows plan)."""
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.projections import create_projection_map
    from tests.helpers import RTOL, SINGLE_MAPS, load, problem, relerr, torch_args

    monkeypatch.setenv(*switch)
    TD = {"f32": torch.float32, "f64": torch.float64}
    for fixture in ("g1_syn2000.npz", "g1_long.npz"):
        z = load(fixture)
        p = problem(z)
        objs = {}
        for key in z["cases"]:
            mk, g, ln, dn = str(key).split("|")
            if mk not in SINGLE_MAPS:
                continue
            if (mk, dn) not in objs:
                pt, pp = SINGLE_MAPS[mk]
                objs[(mk, dn)] = MatchingSolverDualObjectiveFunction(torch_args(p, dn, create_projection_map(pt, dict(pp), p["n"]), DEV), float(g))
                info = objs[(mk, dn)].info()
                if switch[0] == "DUALIP_HIP_SELL":
                    assert info["slices"] == 0, info
                elif switch[0] == "DUALIP_HIP_SELL_LANES":
                    assert info["slice_lane_columns"] == 0, info
            f = objs[(mk, dn)]
            res = f.calculate(torch.from_numpy(z[f"lam_{ln}"]).to(TD[dn]).to(DEV), gamma=float(g), save_primal=True)
            for got, name in ((res.dual_gradient.cpu().numpy(), "grad"), (res.primal_var.cpu().numpy(), "x")):
                assert relerr(got, z[f"{key}|{name}"]) < RTOL[dn], (fixture, key, name)
