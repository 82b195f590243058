"""Callers that keep their tensors on the CPU (``-m gpu``).

The reference's examples default to ``host_device="cpu"`` (examples/movielens_matching/movies_lens_matching.py:227,
examples/miplib_2017/solve_miplib_dataset.py:58) and its own tests build CPU tensors throughout (tests/objectives/*,
tests/projections/*, tests/test_sparse_utils.py).  Under this package the arithmetic only exists in libdualip_hip.so, so such inputs
are STAGED to the current ROCm device and the results handed back on the caller's device (dualip_amd/_hip.py: stage / result_to).
Every case below is the CPU-tensor form of a case the device-tensor tests already pin: same fixtures (reference output), same
tolerances, plus "the answer lives where the question came from".
"""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from tests.helpers import NP_DT, SCALA_GOLDEN, load, problem, relerr, scala_5x5, torch_args

pytestmark = pytest.mark.gpu
TD = {"f32": torch.float32, "f64": torch.float64}


def _all_cpu(res):
    import dataclasses

    for f in dataclasses.fields(res):
        v = getattr(res, f.name)
        if isinstance(v, torch.Tensor):
            assert v.device.type == "cpu", f.name
        elif dataclasses.is_dataclass(v):
            _all_cpu(v)


def test_run_solver_matching_with_host_device_cpu():
    """movies_lens_matching.py:227-275 as it is run by default: ComputeArgs(host_device="cpu").  Trace = fixture G2 (reference)."""
    from dualip_amd.projections import create_projection_map
    from dualip_amd.run_solver import run_solver
    from dualip_amd.types import ComputeArgs, ObjectiveArgs, SolverArgs

    z = load("g2_syn2000.npz")
    p = problem(z)
    args = torch_args(p, "f64", create_projection_map("simplex", {"z": 1.0}, p["n"]), "cpu")
    sa = SolverArgs(max_iter=60, initial_step_size=1e-3, gamma=0.02, max_step_size=1e-1, save_primal=True)
    with contextlib.redirect_stdout(io.StringIO()):
        res = run_solver(args, sa, ComputeArgs(host_device="cpu"), ObjectiveArgs(objective_type="matching"))
    assert relerr(res.dual_objective_log, z["simplex1|f64|dual_obj_log"]) < 1e-6
    _all_cpu(res)
    assert res.objective_result.primal_var is not None and res.objective_result.primal_var.shape == (int(p["colptr"][-1]),)
    assert args.A.values().device.type == "cpu"  # the caller's tensors stay where they were


def test_run_solver_miplib_with_host_device_cpu():
    """solve_miplib_dataset.py:45-75 unchanged: read the shipped .mps.gz, ComputeArgs(host_device="cpu"), 2000 iterations, the driver's
    own check |27 - dual objective| < 1, and the reference's trace (fixture G6) at iterations 100 / 1000 / 2000."""
    from dualip_amd.objectives.miplib import MIPLIBInputArgs
    from dualip_amd.run_solver import run_solver
    from dualip_amd.types import ComputeArgs, ObjectiveArgs, SolverArgs
    from dualip_amd.utils.read_mps_data import read_mps_file

    z = load("g6_miplib_v150.npz")
    data = read_mps_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "v150d30-2hopcds.mps.gz")).to_dualip_format()
    args = MIPLIBInputArgs(A=data.A, c=data.C, b_vec=data.b_vec, projection_map=data.projection_map, equality_mask=data.equality_mask)
    with contextlib.redirect_stdout(io.StringIO()):
        out = run_solver(args, SolverArgs(max_iter=2000, initial_step_size=1e-5, gamma=1e-3), ComputeArgs(host_device="cpu"), ObjectiveArgs(objective_type="miplib2017"))
    want, log = z["trace|f32|obj_log"], np.array(out.dual_objective_log)
    assert relerr(log[:25], want[:25]) < 2e-5 and relerr(log, want) < 2e-2
    assert abs(27 - out.dual_objective) < 1
    assert abs(log[99] - 23.13099) < 0.05 and abs(log[999] - 25.60996) < 0.2 and abs(log[1999] - 27.01548) < 0.3
    _all_cpu(out)


def test_objective_and_maximizer_on_cpu_tensors_scala_known_answers():
    """tests/objectives/test_dualip_matching_simplex.py of the reference, as written there: CPU tensors into the objective, CPU duals
    into calculate() and maximize().  Known answers = the test's own (iteration, dual objective) pairs."""
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent
    from dualip_amd.projections import create_projection_map

    p = scala_5x5()
    args = torch_args(p, "f32", create_projection_map("simplex", {"z": 1.0}, p["n"]), "cpu")
    f = MatchingSolverDualObjectiveFunction(matching_input_args=args, gamma=1e-3)
    assert f.device.type == "cuda"  # staged
    lam = torch.zeros(p["m"], dtype=torch.float32)
    one = f.calculate(lam, gamma=1e-3, save_primal=True)
    _all_cpu(one)
    on_device = f.calculate(lam.to(f.device), gamma=1e-3, save_primal=True)
    assert on_device.dual_gradient.is_cuda and torch.equal(on_device.dual_gradient.cpu(), one.dual_gradient)  # same launch, same bits
    solver = AcceleratedGradientDescent(max_iter=30, gamma=1e-3, iteration_callback=False)
    res = solver.maximize(f, 0.1 * torch.ones(p["m"], dtype=torch.float32))  # (the reference test's start and default step sizes)
    _all_cpu(res)
    for it, want in SCALA_GOLDEN:
        assert abs(res.dual_objective_log[it - 1] - want) < 2e-4 * abs(want), (it, res.dual_objective_log[it - 1], want)


def test_projection_operators_on_cpu_blocks_and_the_private_names():
    """tests/projections/test_{box,cone,simplex}.py call the operators -- and simplex.py's module-level ``_duchi_proj`` /
    ``_proj_via_bisection_search`` -- on CPU blocks, bfloat16 ones included.  Expected values: fixture G-P (reference output)."""
    from dualip_amd.projections.base import project
    from dualip_amd.projections.simplex import _duchi_proj, _proj_via_bisection_search

    x = torch.tensor([[0.5, -0.1], [0.7, 2.0]], dtype=torch.float32)  # (test_simplex.py:6-14)
    w_eq = _duchi_proj(x, z=1.0)
    assert w_eq.device.type == "cpu" and torch.allclose(w_eq.sum(dim=0), torch.tensor([1.0, 1.0]), atol=1e-5) and bool((w_eq >= 0).all())
    w_in = _duchi_proj(x, z=1.0, inequality=True)
    assert bool((w_in.sum(dim=0) <= 1.0 + 1e-5).all()) and bool((w_in >= 0).all())
    xb = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]], dtype=torch.float32).T  # (test_simplex.py:17-40)
    for fn in (_duchi_proj, _proj_via_bisection_search):
        got32, got16 = fn(xb, 1.0), fn(xb.to(torch.bfloat16), 1.0)
        assert got16.dtype == torch.bfloat16 and got16.device.type == "cpu"
        assert torch.allclose(got32.sum(dim=0), torch.tensor(1.0), atol=1e-5)
        assert torch.allclose(got16.to(torch.float32).sum(dim=0), torch.tensor(1.0), atol=1e-2)
    rng = np.random.default_rng(0)  # duchi and bisection agree where both are projections (test_simplex.py:125-160)
    blk = torch.from_numpy(rng.uniform(0.0, 2.0, (17, 40))).float()
    assert torch.allclose(_duchi_proj(blk, 1.0), _proj_via_bisection_search(blk, 1.0), atol=1e-4)
    z = load("gp_projections.npz")
    ops = {"simplex_z1": ("simplex", {"z": 1.0}), "simplex_eq_z1": ("simplex_eq", {"z": 1.0}), "simplex_bisect_z1": ("simplex", {"z": 1.0, "method": "bisection_search"}),
           "box": ("box", {"lower": -0.2, "upper": 0.7}), "cone_lo": ("cone", {"lower": 0.1}), "cone_up": ("cone", {"upper": 0.1})}
    for bn in z["blocks"]:
        for on, (pt, pp) in ops.items():
            for dn in NP_DT:
                xin = torch.from_numpy(z[f"in|{bn}"]).to(TD[dn])
                got = project(pt, **pp)(xin)
                assert got.device.type == "cpu" and got.dtype == TD[dn]
                tol = (1e-9 if dn == "f64" else 5e-6) if "bisect" in on else (1e-12 if dn == "f64" else 2e-6)
                assert np.allclose(got.numpy(), z[f"out|{bn}|{on}|{dn}"], rtol=0, atol=tol), (bn, on, dn)


def test_sparse_utils_on_cpu_tensors_including_output_tensor():
    """tests/test_sparse_utils.py of the reference works on CPU CSC tensors and checks ``output_tensor`` written in place."""
    from dualip_amd.preprocessing.precondition import jacobi_precondition
    from dualip_amd.projections.base import project
    from dualip_amd.utils.sparse_utils import apply_F_to_columns, elementwise_csc, left_multiply_sparse, right_multiply_sparse, row_norms_csc, row_sums_csc

    rng = np.random.default_rng(5)
    dense = rng.uniform(-1.0, 1.0, (7, 9)) * (rng.uniform(size=(7, 9)) < 0.5)
    dense[:, 4] = 0.0
    A = torch.from_numpy(dense).to_sparse_csc()
    B = torch.sparse_csc_tensor(A.ccol_indices(), A.row_indices(), torch.from_numpy(rng.uniform(0.5, 1.5, A.values().shape[0])), size=A.shape)
    v_rows, v_cols = torch.from_numpy(rng.uniform(0.5, 2.0, 7)), torch.from_numpy(rng.uniform(0.5, 2.0, 9))
    assert torch.allclose(left_multiply_sparse(v_rows, A).to_dense(), torch.diag(v_rows) @ A.to_dense(), atol=1e-14)
    assert torch.allclose(right_multiply_sparse(A, v_cols).to_dense(), A.to_dense() @ torch.diag(v_cols), atol=1e-14)
    assert torch.allclose(row_sums_csc(A), A.to_dense().sum(dim=1), atol=1e-13) and row_sums_csc(A).device.type == "cpu"
    assert torch.allclose(row_norms_csc(A), A.to_dense().norm(dim=1), atol=1e-13)
    assert torch.allclose(elementwise_csc(A, B, torch.mul).to_dense(), A.to_dense() * B.to_dense(), atol=1e-14)
    out = torch.sparse_csc_tensor(A.ccol_indices(), A.row_indices(), torch.zeros_like(A.values()), size=A.shape)
    ret = left_multiply_sparse(v_rows, A, output_tensor=out)
    assert ret.data_ptr() == out.values().data_ptr() and torch.allclose(out.to_dense(), torch.diag(v_rows) @ A.to_dense(), atol=1e-14)
    P = apply_F_to_columns(B, project("simplex", z=1.0), [torch.arange(0, 5), torch.arange(5, 9)])
    assert P.values().device.type == "cpu"
    Pd = P.to_dense()
    assert bool((Pd >= 0).all()) and bool((Pd.sum(dim=0) <= 1.0 + 1e-9).all())
    # jacobi_precondition scales A and b IN PLACE (preprocessing/precondition.py:8-28) -- the CPU tensors themselves
    A2 = torch.sparse_csc_tensor(A.ccol_indices(), A.row_indices(), A.values().clone(), size=A.shape)
    b = torch.from_numpy(rng.uniform(1.0, 2.0, 7))
    b0 = b.clone()
    norms = jacobi_precondition(A2, b)
    want = A.to_dense().norm(dim=1)
    want = torch.where(want == 0, torch.ones_like(want), want)
    assert norms.device.type == "cpu" and torch.allclose(norms, want, atol=1e-13)
    assert torch.allclose(A2.to_dense(), A.to_dense() / want[:, None], atol=1e-13) and torch.allclose(b, b0 / want, atol=1e-13)


def test_host_arrays_are_staged_natively_and_give_the_same_bits():
    """``dl_stage_to_device`` (pinned, chunked, several DMA queues; int64 row indices narrowed on the host): a plain copy arrives bit for bit,
    narrowing int64 -> uint16 / int32 gives the values, a value that does not fit is an error; an objective built from CPU tensors in the
    reference's format (two int64 CSC tensors of one pattern) returns the SAME BITS as one built from device tensors -- ``calculate``, primal
    and a 40-iteration solve -- keeps the caller's tensors as ``objective.A`` / ``.c``, and follows in-place changes of the host arrays
    through ``values_changed()``."""
    from dualip_amd import _hip
    from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent
    from dualip_amd.projections import create_projection_map

    if os.environ.get("DUALIP_HOST_STAGING") is not None:
        pytest.skip("states the default staging route (the suite is also run with DUALIP_HOST_STAGING=torch set for every test)")
    g = torch.Generator().manual_seed(5)
    for n_el in (1, 1000, (16 << 20) // 4 + 17, 3 * (16 << 20) // 4 + 5):  # (below one chunk, exactly past one, several chunks with a ragged tail)
        src = torch.randn(n_el, generator=g)
        assert torch.equal(_hip.stage_array(src, "cuda:0").cpu(), src)
    idx = torch.randint(0, 65536, (5_000_003,), generator=g, dtype=torch.int64)
    u16 = _hip.stage_array(idx, "cuda:0", narrow_to=torch.uint16)
    assert u16.dtype == torch.int16 and torch.equal(u16.cpu().to(torch.int64) & 0xFFFF, idx)
    assert torch.equal(_hip.stage_array(idx, "cuda:0", narrow_to=torch.int32).cpu().to(torch.int64), idx)
    idx[1234567] = 65536
    with pytest.raises(ValueError, match="do not fit"):
        _hip.stage_array(idx, "cuda:0", narrow_to=torch.uint16)
    idx[1234567] = -1
    with pytest.raises(ValueError, match="do not fit"):
        _hip.stage_array(idx, "cuda:0", narrow_to=torch.uint16)

    z = load("g2_syn2000.npz")
    p = problem(z)
    n = p["n"]
    half = n // 2
    pm = {**create_projection_map("box", {"lower": 0.0, "upper": 1.0}, None, indices=range(0, half)), **create_projection_map("simplex", {"z": 1.0}, None, indices=range(half, n))}
    kw = dict(max_iter=40, gamma=0.02, initial_step_size=1e-3, max_step_size=0.1, iteration_callback=False)
    lam = torch.from_numpy(np.random.default_rng(2).uniform(0, 0.02, p["m"]))
    for dn in ("f32", "f64"):
        dev_args, cpu_args = torch_args(p, dn, pm, "cuda:0"), torch_args(p, dn, pm, "cpu")
        fd = MatchingSolverDualObjectiveFunction(dev_args, gamma=0.02)
        before = len(_hip.STAGING_LOG)
        fc = MatchingSolverDualObjectiveFunction(cpu_args, gamma=0.02)
        staged = _hip.STAGING_LOG[before:]
        assert [r["what"] for r in staged] == ["ccol_indices", "row_indices", "A.values", "c.values"], staged
        assert staged[1]["bytes_link"] * 4 == staged[1]["bytes_host"]  # int64 row indices crossed the link as 16 bits (m <= 65536)
        assert fc.A is cpu_args.A and fc.c is cpu_args.c and fc.A.values().device.type == "cpu" and fc.device.type == "cuda"
        rd = fd.calculate(lam.to(TD[dn]).to("cuda:0"), save_primal=True)
        rc = fc.calculate(lam.to(TD[dn]), save_primal=True)  # (CPU duals in, CPU results out)
        assert rc.dual_gradient.device.type == "cpu" and torch.equal(rc.dual_gradient, rd.dual_gradient.cpu()) and torch.equal(rc.primal_var, rd.primal_var.cpu())
        assert float(rc.dual_objective) == float(rd.dual_objective)
        sd = AcceleratedGradientDescent(**kw).maximize(fd, torch.zeros(p["m"], dtype=TD[dn], device="cuda:0"))
        sc = AcceleratedGradientDescent(**kw).maximize(fc, torch.zeros(p["m"], dtype=TD[dn]))
        assert sc.dual_objective_log == sd.dual_objective_log and torch.equal(sc.dual_val, sd.dual_val.cpu())
        # an in-place change of the caller's HOST arrays reaches the handle through values_changed(), as for device callers
        cpu_args.A.values().mul_(0.5)
        dev_args.A.values().mul_(0.5)
        fc.values_changed()
        fd.values_changed()
        assert torch.equal(fc.calculate(lam.to(TD[dn])).dual_gradient, fd.calculate(lam.to(TD[dn]).to("cuda:0")).dual_gradient.cpu())
    # torch's own copy stays available (and is what maps with user-defined operators / Jacobi preconditioning take)
    os.environ["DUALIP_HOST_STAGING"] = "torch"
    try:
        before = len(_hip.STAGING_LOG)
        ft = MatchingSolverDualObjectiveFunction(torch_args(p, "f64", pm, "cpu"), gamma=0.02)
        assert len(_hip.STAGING_LOG) == before and ft.A.values().is_cuda
    finally:
        del os.environ["DUALIP_HOST_STAGING"]
