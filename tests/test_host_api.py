"""CPU tests of the host-side API surface (no GPU): registry / map builder / sharding helpers / generic maximiser /
C-ABI export list.  The cases mirror the reference's own tests (cited per test) with this package's import root."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from dualip_amd import _hip
from dualip_amd.optimizers.agd import AcceleratedGradientDescent, compute_beta_seq, project_on_nn_cone
from dualip_amd.optimizers.agd_utils import (
    calculate_step_size,
    estimate_lipschitz_constant,
    norm_of_difference,
    step_size_from_lipschitz_constants,
    update_dual_gradient_history,
)
from dualip_amd.projections import ProjectionEntry, create_projection_map, project
from dualip_amd.types import ObjectiveResult
from dualip_amd.utils.dist_utils import global_to_local_projection_map, split_tensors_to_devices
from dualip_amd.utils.sparse_utils import hstack_csc, split_csc_by_cols
from tests.helpers import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- C ABI ------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "dualip_hip.h")).read()
    declared = set(re.findall(r"\b(dl_[a-z0-9_]+)\s*\(", header))
    declared -= {"dl_stream_t"}
    assert len(declared) >= 18
    lib = ctypes.CDLL(_hip._build.build())  # loading needs no GPU; no compute entry point is called
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/dualip_hip.h but not exported"
    assert declared == set(_hip._SIGNATURES), "python binding table and header disagree"
    lib.dl_version.restype = ctypes.c_int
    assert lib.dl_version() >= 100


def test_cpu_tensors_are_rejected_loudly():
    from dualip_amd.objectives.matching import MatchingInputArgs, MatchingSolverDualObjectiveFunction

    a = torch.eye(3).to_sparse_csc()
    args = MatchingInputArgs(A=a, c=a, projection_map=create_projection_map("box", {}, 3), b_vec=torch.ones(3))
    with pytest.raises(_hip.HipLibraryError, match="no CPU fallback"):
        MatchingSolverDualObjectiveFunction(args, gamma=1e-3)
    with pytest.raises(_hip.HipLibraryError):
        project("box")(torch.zeros(3, 2))


# ---- projections registry (reference projections/base.py:40-97) ---------------------------------------------
def test_projection_map_keys_and_registry():
    pm = create_projection_map("simplex", {"z": 1.0}, 5)
    assert list(pm) == ["simplex_z_1.0"] and pm["simplex_z_1.0"].indices == [0, 1, 2, 3, 4]
    pm = create_projection_map("box", {"upper": 1.0, "lower": 0.0}, 10, indices=[0, 2, 4], key_prefix="p_")
    assert list(pm) == ["p_box_lower_0.0_upper_1.0"] and pm["p_box_lower_0.0_upper_1.0"].indices == [0, 2, 4]
    assert isinstance(create_projection_map("box", {}, 5_000_000)["box_"].indices, range)
    with pytest.raises(ValueError, match="Unknown projection operator 'nope'"):
        project("nope")
    with pytest.raises(ValueError, match="Only one of"):
        project("cone", lower=0.0, upper=1.0)
    with pytest.raises(ValueError, match="Unsupported projection method"):
        project("simplex", z=1.0, method="sorting")
    d = project("box", lower=0.25).descriptor()
    assert (d.kind, d.p0, d.p1) == (_hip.PROJ_BOX, 0.25, 1.0)
    assert project("cone").descriptor().kind == _hip.PROJ_NONE
    assert project("cone", upper=2.0).descriptor().kind == _hip.PROJ_CONE_UPPER
    assert project("simplex_eq", z=2.0).descriptor().kind == _hip.PROJ_SIMPLEX_EQ
    assert ProjectionEntry().indices == []


# ---- sharding helpers (reference tests/test_dist_utils.py:8-97) ---------------------------------------------
def test_global_to_local_projection_map():
    a = torch.randn(5, 6).to_sparse_csc()
    pm = create_projection_map("simplex_ineq", {"z": 1}, 6)
    _, _, index_map = split_tensors_to_devices(a, a, ["cpu", "cpu"])
    local = [global_to_local_projection_map(pm, cols) for cols in index_map]
    assert local[0]["simplex_ineq_z_1"].indices == [0, 1, 2]
    assert local[1]["simplex_ineq_z_1"].indices == [0, 1, 2]
    pm = {**create_projection_map("simplex_ineq", {"z": 1}, 10, indices=[0, 1]), **create_projection_map("simplex_eq", {"z": 2}, 10, indices=[2, 3, 4, 5, 6, 7, 8, 9])}
    a = torch.randn(10, 10).to_sparse_csc()
    _, _, index_map = split_tensors_to_devices(a, a, ["cpu", "cpu"])
    local = [global_to_local_projection_map(pm, cols) for cols in index_map]
    assert local[0]["simplex_ineq_z_1"].indices == [0, 1]
    assert local[0]["simplex_eq_z_2"].indices == [2, 3, 4]
    assert local[1]["simplex_eq_z_2"].indices == [0, 1, 2, 3, 4]
    assert "simplex_ineq_z_1" not in local[1]
    # explicit (non-range) column lists still work
    assert global_to_local_projection_map(pm, [5, 6, 7, 8, 9])["simplex_eq_z_2"].indices == [0, 1, 2, 3, 4]


def test_split_tensors_to_devices():
    a, c = torch.randn(5, 6).to_sparse_csc(), torch.randn(5, 6).to_sparse_csc()
    a_s, c_s, _ = split_tensors_to_devices(a, c, ["cpu", "cpu"])
    assert [t.shape for t in a_s] == [(5, 3), (5, 3)] and [t.shape for t in c_s] == [(5, 3), (5, 3)]
    a, c = torch.randn(5, 5).to_sparse_csc(), torch.randn(5, 5).to_sparse_csc()
    a_s, c_s, imap = split_tensors_to_devices(a, c, ["cpu", "cpu"])
    assert [t.shape for t in a_s] == [(5, 3), (5, 2)] and list(imap[1]) == [3, 4]
    a_s, c_s, imap = split_tensors_to_devices(a, c, [])
    assert len(a_s) == 1 and torch.equal(a_s[0].values(), a.values()) and imap == [0, 1, 2, 3, 4]
    a = torch.randn(5, 7).to_sparse_csc()
    a_s, _, _ = split_tensors_to_devices(a, a, ["cpu"] * 3)
    assert [t.shape[1] for t in a_s] == [3, 2, 2]
    # blocks re-assemble to the original matrix, column pointers are re-based
    assert torch.equal(hstack_csc(a_s).to_dense(), a.to_dense())
    assert all(int(t.ccol_indices()[0]) == 0 for t in a_s)
    with pytest.raises(ValueError):
        split_csc_by_cols(a, [3, 3])
    with pytest.raises(ValueError):
        split_tensors_to_devices(a.to_dense(), a, ["cpu"])


# ---- step-size helpers (reference tests/test_utils.py:12-96) -------------------------------------------------
def test_step_size_helpers():
    assert torch.allclose(norm_of_difference(torch.tensor([1.0, 2.0, 3.0]), torch.tensor([4.0, 5.0, 6.0])), torch.sqrt(torch.tensor(27.0)))
    gh, dh = [], []
    for k in range(3):
        update_dual_gradient_history(torch.tensor([1.0 + 4 * k, 2.0 + 4 * k]), torch.tensor([3.0 + 4 * k, 4.0 + 4 * k]), gh, dh, 2)
    assert len(gh) == 2 and len(dh) == 2 and torch.allclose(gh[0], torch.tensor([5.0, 6.0])) and torch.allclose(gh[1], torch.tensor([9.0, 10.0]))
    L = estimate_lipschitz_constant(torch.tensor([1.0, 2.0]), torch.tensor([3.0, 4.0]), torch.tensor([5.0, 6.0]), torch.tensor([7.0, 8.0]))
    assert isinstance(L, torch.Tensor) and L > 0
    assert step_size_from_lipschitz_constants([], 5, 0.1, 1.0) == 0.1
    assert step_size_from_lipschitz_constants([torch.tensor(1.0), torch.tensor(2.0)], 5, 0.1, 1.0) == 0.1
    assert step_size_from_lipschitz_constants([torch.tensor(1.0)] * 5, 5, 0.1, 1.0) == 1.0
    assert step_size_from_lipschitz_constants([torch.tensor(float("nan"))] * 5, 5, 0.1, 1.0) == 0.1
    assert step_size_from_lipschitz_constants([torch.tensor(0.0)] * 4, 5, 0.1, 0.7) == 0.7
    # a NaN that is not first is skipped by builtins.max
    mixed = [torch.tensor(2.0), torch.tensor(float("nan")), torch.tensor(4.0), torch.tensor(1.0)]
    assert step_size_from_lipschitz_constants(mixed, 5, 0.1, 1.0) == 0.25
    s = calculate_step_size(torch.tensor([1.0, 2.0]), torch.tensor([3.0, 4.0]), [], [], max_history_length=5, initial_step_size=0.1, max_step_size=1.0)
    assert isinstance(s, float) and s == 0.1


def test_project_on_nn_cone():
    # reference tests/test_equality_constraints.py:8-15
    y = torch.tensor([-1.0, -1.0, 2.0, -3.0, 4.0])
    mask = torch.tensor([False, True, False, True, False])
    assert torch.equal(project_on_nn_cone(y, mask), torch.tensor([0.0, -1.0, 2.0, -3.0, 4.0]))
    assert torch.equal(project_on_nn_cone(y), torch.tensor([0.0, 0.0, 2.0, 0.0, 4.0]))


def test_beta_seq_bit_exact_with_reference():
    want = load("g4_beta_seq.npz")["beta"]
    got = compute_beta_seq(want.shape[0]).numpy()
    assert got.dtype == np.float32 and np.array_equal(got, want)
    assert compute_beta_seq(0).numel() == 0


# ---- generic maximiser on user-defined objectives (reference tests/test_agd.py:48-109) ---------------------
class _Quadratic1D:
    equality_mask = None

    def calculate(self, dual_val, save_primal=False, **kwargs):
        x = dual_val[0]
        return ObjectiveResult(dual_gradient=torch.tensor([-2.0 * (x - 3.0)]), dual_objective=-((x - 3.0) ** 2), reg_penalty=None)


class _Quadratic2D:
    equality_mask = None

    def calculate(self, dual_val, save_primal=False, **kwargs):
        x, y = dual_val
        return ObjectiveResult(dual_gradient=torch.tensor([-2.0 * (x - 3.0), -2.0 * (y + 5.0)]), dual_objective=-((x - 3.0) ** 2) - (y + 5.0) ** 2, reg_penalty=None)


def test_generic_maximizer_first_step():
    for step in (1e-5, 0.1):
        s = AcceleratedGradientDescent(max_iter=1, gamma=None, initial_step_size=step, iteration_callback=False)
        r = s.maximize(_Quadratic1D(), torch.tensor([0.0]))
        assert abs(r.dual_val[0] - 6.0 * step) < 1e-10


def test_generic_maximizer_known_answer_trace(capsys):
    s = AcceleratedGradientDescent(max_iter=30, gamma=None, initial_step_size=1e-5)
    r = s.maximize(_Quadratic2D(), torch.tensor([0.0, 0.0]))
    for i, want in [(2, -33.9996400036), (16, -28.60551547593112), (23, -25.473701313626133), (29, -25.00382134903756)]:
        assert abs(r.dual_objective_log[i - 1] - want) < 1e-5
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 30 and out[0].startswith("iter=1 | dual_objective=") and "dual_grad_norm=" in out[0]
    with pytest.raises(ValueError, match="Unsupported gamma decay type"):
        AcceleratedGradientDescent(max_iter=1, gamma=1.0, gamma_decay_type="exp", gamma_decay_params={}, iteration_callback=False).maximize(
            _Quadratic1D(), torch.tensor([0.0])
        )


def test_run_solver_rejects_unknown_objective():
    from dualip_amd.run_solver import build_objective
    from dualip_amd.types import ComputeArgs, ObjectiveArgs, SolverArgs

    with pytest.raises(ValueError, match="not supported"):
        build_objective(None, SolverArgs(), ComputeArgs(host_device="cpu"), ObjectiveArgs(objective_type="other"))


def test_balanced_block_ranges_cover_every_block_evenly():
    from dualip_amd.utils.dist_utils import balanced_block_ranges

    blocks = [(0, 1000), (1000, 2600)]
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            rs = balanced_block_ranges(blocks, world, r, align=100)
            for a, b in rs:
                assert a % 100 == 0 and (b % 100 == 0 or b in (1000, 2600))
            seen.append(rs)
            # every rank holds a share of both blocks (10 and 16 units of 100 over <= 8 ranks)
            assert any(a < 1000 for a, _ in rs) and any(a >= 1000 for a, _ in rs)
        cols = sorted(c for rs in seen for a, b in rs for c in range(a, b))
        assert cols == list(range(2600))
        per_block = [[sum(b - a for a, b in rs if lo <= a < hi) for rs in seen] for lo, hi in blocks]
        for sizes in per_block:
            assert max(sizes) - min(sizes) <= 100
    # unaligned tail and more parts than units
    rs = [balanced_block_ranges([(0, 250)], 4, r, align=100) for r in range(4)]
    assert sorted(c for x in rs for a, b in x for c in range(a, b)) == list(range(250))


def test_memmap_cache_format_reads_and_rewrites_the_reference_cache(tmp_path):
    """tests/golden/g5_cache/ was written by the reference's generator (tests/golden/make_golden_cache.py)."""
    import filecmp
    import json
    import os

    import numpy as np
    import torch

    from benchmark import cache_format as cf
    from tests.helpers import GOLDEN

    src = os.path.join(GOLDEN, "g5_cache")
    key = (1000, 20, 0.2, torch.float32, 42)
    assert cf.cache_prefix(*key) == "s1000_d20_sp0.2_float32_seed42"
    arrays = cf.load_matching_cache_numpy(src, *key)
    assert arrays is not None
    ccol, row, a_vals, c_vals, b = arrays
    assert ccol.shape == (1001,) and ccol[0] == 0 and ccol[-1] == row.shape[0] == a_vals.shape[0] == c_vals.shape[0] == 3993
    assert np.all(np.diff(ccol) >= 0) and row.min() >= 0 and row.max() < 20 and b.shape == (20,)
    assert np.all(c_vals >= 0) and np.all(a_vals >= 0)  # the cache holds positive costs; the input bundle negates them (:447-448)
    # a different key is a miss, as in the reference (generate_synthetic_data.py:235-246)
    assert cf.load_matching_cache_numpy(src, 1000, 20, 0.2, torch.float32, 43) is None
    assert cf.load_matching_cache_numpy(src, 1000, 20, 0.2, torch.float64, 42) is None
    # writing the same arrays gives byte-identical files and the same metadata record
    dst = str(tmp_path / "cache")
    prefix = cf.save_matching_cache(dst, *key, ccol, row, a_vals, c_vals, b)
    for k in cf.ARRAYS:
        assert filecmp.cmp(os.path.join(src, f"{prefix}_{k}.dat"), os.path.join(dst, f"{prefix}_{k}.dat"), shallow=False), k
    assert json.load(open(os.path.join(src, f"{prefix}_meta.json"))) == json.load(open(os.path.join(dst, f"{prefix}_meta.json")))
    # and loads into the operator API's input bundle (CPU tensors here; the GPU tests load to the device)
    args = cf.load_matching_cache(src, *key, device="cpu")
    assert args.A.layout == torch.sparse_csc and tuple(args.A.shape) == (20, 1000) and args.A.values().dtype == torch.float32
    assert torch.equal(args.A.ccol_indices(), args.c.ccol_indices()) and args.b_vec.shape == (20,)
    assert torch.equal(args.c.values(), -torch.from_numpy(np.asarray(c_vals)).to(torch.float32)) and "simplex_z_1.0" in args.projection_map
    # bundle -> cache -> bundle round trip
    dst2 = str(tmp_path / "cache2")
    cf.save_matching_args(dst2, args, 0.2, 7)
    back = cf.load_matching_cache(dst2, 1000, 20, 0.2, torch.float32, 7, device="cpu")
    assert torch.equal(back.A.values(), args.A.values()) and torch.equal(back.c.values(), args.c.values()) and torch.equal(back.b_vec, args.b_vec)


def _read_metrics(run_dir):
    import csv

    with open(os.path.join(run_dir, "metrics.csv")) as fh:
        rows = list(csv.DictReader(fh))
    by_key = {}
    for r in rows:
        by_key.setdefault(r["key"], []).append((int(r["step"]), float(r["value"])))
    return by_key


def test_run_tracking_file_store(tmp_path):
    """Tracking (reference: utils/mlflow_utils.py) with the file store that stands in when mlflow is not installed: the
    generic AGD route logs step_size / dual_objective / gamma and the objective's scalars every iteration (agd.py:189-201),
    hyper-parameters go to params.json, nothing is written outside a run context or with enabled=False, and a broken
    backend never stops the solve."""
    import json

    from dualip_amd.utils import mlflow_utils as mu

    if mu.is_mlflow_available():
        pytest.skip("mlflow installed: the file store is not used")
    cfg = mu.MLflowConfig(enabled=True, tracking_uri=str(tmp_path), experiment_name="exp", run_name="quad")
    solver = AcceleratedGradientDescent(max_iter=20, gamma=0.5, initial_step_size=1e-3, gamma_decay_type="step", gamma_decay_params={"decay_steps": 8, "decay_factor": 0.5}, iteration_callback=False)
    with mu.mlflow_run_context(cfg) as run:
        assert run == os.path.join(str(tmp_path), "exp", "quad") and mu.tracking_enabled()
        mu.log_hyperparameters({"solver": {"max_iter": 20, "gamma": 0.5, "save_primal": False, "gamma_decay_type": None}, "objective": {"objective_type": "matching", "objective_kwargs": {}}})
        res = solver.maximize(_Quadratic2D(), torch.tensor([0.0, 0.0]))
    assert not mu.tracking_enabled()
    with open(os.path.join(run, "params.json")) as fh:
        assert json.load(fh) == {"solver.max_iter": 20, "solver.gamma": 0.5, "solver.gamma_decay_type": "None", "objective.objective_type": "matching"}
    got = _read_metrics(run)
    assert [s for s, _ in got["step_size"]] == list(range(1, 21))
    assert [v for _, v in got["step_size"]] == res.step_size_log
    # the per-iteration record and the objective record both carry the dual objective (as the reference)
    assert [v for s, v in got["dual_objective"]][0::2] == pytest.approx(res.dual_objective_log, rel=1e-15)
    assert [v for _, v in got["gamma"]] == [0.5 * 0.5 ** (i // 8) for i in range(1, 21)]
    assert "regularization_penalty" not in got  # the toy objective reports none
    # disabled config: no directory, no state
    with mu.mlflow_run_context(mu.MLflowConfig(enabled=False, tracking_uri=str(tmp_path / "off"))) as run2:
        assert run2 is None and not mu.tracking_enabled()
        mu.log_metrics({"a": 1.0}, step=1)
    assert not (tmp_path / "off").exists()
    # a second run of the same name gets its own directory
    with mu.mlflow_run_context(cfg) as run3:
        mu.log_metrics({"a": 1.0, "skipped": "text"}, step=3)
    assert run3.endswith("quad_1") and _read_metrics(run3) == {"a": [(3, 1.0)]}
    # an unusable location is reported, not raised
    blocker = tmp_path / "file"
    blocker.write_text("x")
    with mu.mlflow_run_context(mu.MLflowConfig(enabled=True, tracking_uri=str(blocker))) as run4:
        assert run4 is None and not mu.tracking_enabled()


def test_user_defined_projection_operator_registers_without_a_kernel_form():
    from dualip_amd.projections.base import ProjectionOperator, register

    @register("user_halve")
    class Halve(ProjectionOperator):
        def __init__(self, factor=0.5):
            self.factor = factor

        def __call__(self, x):
            return x * self.factor

    op = project("user_halve", factor=0.25)
    assert op.descriptor() is None and torch.equal(op(torch.tensor([4.0, 8.0])), torch.tensor([1.0, 2.0]))


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/dualip_hip.h is consumable by a C compiler (the boundary has no C++ or torch types), and a C translation
    unit that only includes it links against the shared object and can call an entry point that needs no GPU."""
    import shutil
    import subprocess

    from dualip_amd import _build

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = os.path.join(root, "include", "dualip_hip.h")
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", header], check=True)
    src = tmp_path / "use.c"
    src.write_text('#include <stdio.h>\n#include "dualip_hip.h"\nint main(void) { printf("%d\\n", dl_version()); return dl_last_error_string() == 0; }\n')
    exe = tmp_path / "use"
    lib_dir = os.path.dirname(_build.build())
    r = subprocess.run([gcc, "-std=c99", "-I", os.path.dirname(header), str(src), "-o", str(exe), "-L", lib_dir, "-ldualip_hip", f"-Wl,-rpath,{lib_dir}", "-Wl,--allow-shlib-undefined"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot link against the HIP runtime here: " + r.stderr[-300:])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    if out.returncode == 0:
        assert int(out.stdout.strip()) >= 1


def test_build_screens_device_assembly_for_the_misplaced_spill():
    """Regression guard for the build-dependent wrong result of round 2: hipcc placed a VGPR spill ahead of
    the exec restore of a control-flow join, so wavefronts that skipped the branch reloaded garbage.  The in-tree build assembles
    every object from device assembly it has screened for that pattern; the screen must flag the offending excerpt (kept as a
    fixture, compiler output of this repository's own kernel) and pass its corrected form."""
    from dualip_amd import _build

    bad = os.path.join(ROOT, "tests", "golden", "spill_before_exec_restore.s")
    found = _build._spill_defects(bad)
    assert len(found) == 1 and "scratch_store_dword off, v70" in found[0] and "matching_fused_kernel4IfjLb0ELb0" in found[0], found
    text = open(bad).read()
    # the same block with the spill AFTER the exec restore (what a correct allocation looks like) is clean
    fixed = text.replace("\tscratch_store_dword off, v70, off offset:24 ; 4-byte Folded Spill\n\ts_nop 0\n", "").replace(
        "\ts_or_b64 exec, exec, s[2:3]\n", "\ts_or_b64 exec, exec, s[2:3]\n\tscratch_store_dword off, v70, off offset:24 ; 4-byte Folded Spill\n")
    assert fixed != text
    import tempfile

    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as fh:
        fh.write(fixed)
    try:
        assert _build._spill_defects(fh.name) == []
    finally:
        os.unlink(fh.name)
    # SGPR spills (v_writelane ignores exec) in the same place are legitimate and must not be flagged
    only_sgpr = text.replace("\tscratch_store_dword off, v70, off offset:24 ; 4-byte Folded Spill\n", "")
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as fh:
        fh.write(only_sgpr)
    try:
        assert _build._spill_defects(fh.name) == []
    finally:
        os.unlink(fh.name)


def test_build_screens_device_assembly_for_the_half_redefined_scalar_pair():
    """Guard for the SECOND code-generation defect: a 64-bit scalar value whose high half was overwritten by
    an unrelated scalar load while the pair was live, spilled to VGPR lanes and reloaded as a pair.  The screen must flag the
    hand-written listing, and pass (a) the same code with gridDim.x loaded into a register OUTSIDE the pair, (b) a zero-extension
    (the high half redefined by s_mov, a legitimate way to form a 64-bit value), (c) two unrelated 32-bit values that merely sit
    in adjacent lanes and are never used as a pair."""
    import tempfile

    from dualip_amd import _build

    bad = os.path.join(ROOT, "tests", "golden", "sgpr_pair_half_redefined.s")
    found = _build._sgpr_pair_defects(bad)
    assert len(found) == 1 and "s[10:11]" in found[0] and "s55" in found[0] and "lanes 34/35" in found[0] and "matching_fused_kernel4IdtLb1ELb1ELb0ELb0ELb1" in found[0], found
    assert _build._spill_defects(bad) == []  # (the other screen has nothing to say about it)
    text = open(bad).read()
    variants = {
        "other register": text.replace("s_load_dword s55, s[0:1], 0x230", "s_load_dword s56, s[0:1], 0x230").replace("s_mul_i32 s4, s55, 16", "s_mul_i32 s4, s56, 16"),
        "zero extension": text.replace("s_load_dword s55, s[0:1], 0x230", "s_mov_b32 s55, 0"),
        "never used as a pair": text.replace("s_lshl_b64 s[8:9], s[10:11], 3", "s_lshl_b32 s8, s10, 3"),
        # (a false positive of the first version, met in a real build: the pair was re-formed as a lane mask by a VECTOR compare before the spill)
        "pair re-formed by a vector compare": text.replace("\ts_mul_i32 s4, s55, 16\n", "\ts_mul_i32 s4, s55, 16\n\tv_cmp_lt_i32_e64 s[54:55], 27, v79\n"),
    }
    for name, var in variants.items():
        assert var != text, name
        with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as fh:
            fh.write(var)
        try:
            assert _build._sgpr_pair_defects(fh.name) == [], name
        finally:
            os.unlink(fh.name)
    # after the reload: redefining the STALE half (s11, reloaded from s55's lane) heals the pair; redefining the HEALTHY half (s10) does not --
    # the stale one still rides into the 64-bit operand (round-5 review: the screen used to drop the report on either)
    still_bad = {
        "healthy half redefined after the reload": text.replace("\ts_mul_i32 s5, s11, s6\n", "\ts_mul_i32 s5, s11, s6\n\ts_add_i32 s10, s10, 1\n"),
    }
    healed = {
        "stale half redefined after the reload": text.replace("\ts_mul_i32 s5, s11, s6\n", "\ts_mul_i32 s5, s11, s6\n\ts_mov_b32 s11, 0\n"),
    }
    for group, want_findings in ((still_bad, 1), (healed, 0)):
        for name, var in group.items():
            assert var != text, name
            with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as fh:
                fh.write(var)
            try:
                assert len(_build._sgpr_pair_defects(fh.name)) == want_findings, name
            finally:
                os.unlink(fh.name)


def test_build_manifest_records_what_the_screens_saw():
    """dualip_amd/lib/build_manifest.json (written by every build): per translation unit the number of device-assembly files the
    screens read (never zero: a build must not pass because there was nothing to look at), their findings (none), and per kernel
    the registers / spills / scratch of the code object's metadata.  The benchmark's instantiation of the fused kernel must not
    touch scratch."""
    import json

    from dualip_amd import _build

    _build.build()
    man = json.load(open(_build.MANIFEST_PATH))
    assert sorted(o["source"] for o in man["objects"]) == sorted(_build.SOURCES)
    kernels = {}
    for o in man["objects"]:
        assert o["screened_assembly_files"] >= 1, o["source"]
        assert o["spill_before_exec_restore"] == [] and o["sgpr_pair_half_redefined"] == [], o["source"]
        kernels.update(o["kernels"])
    bench = kernels[man["benchmark_kernel"]]
    assert man["benchmark_kernel"] == _build.BENCHMARK_KERNEL and "matching_fused_kernel4IftLb1ELb1ELb0ELb0ELb0" in man["benchmark_kernel"]
    assert bench["vgpr_spill_count"] == 0 and bench["scratch_bytes"] == 0 and bench["vgpr_count"] <= 128, bench
    assert len(kernels) > 60 and all(v["vgpr_count"] is not None for v in kernels.values())
    # the fused kernels ask for all 160 KB of LDS dynamically (hipFuncAttributeMaxDynamicSharedMemorySize = kLdsBudget): a single byte of STATIC
    # LDS -- a __shared__ array, or a builtin that brings its own (__syncthreads_or: 256 bytes) -- makes that request invalid and every handle
    # creation fail at run time (round 5 found out on the GPU box)
    fused = {k: v for k, v in kernels.items() if "matching_fused_kernel" in k}
    assert len(fused) >= 20 and all(v["lds_bytes"] == 0 for v in fused.values()), {k: v["lds_bytes"] for k, v in fused.items() if v["lds_bytes"]}


def test_build_lists_cover_every_source_and_header():
    """The build hashes SOURCES + HEADERS to decide staleness and compiles exactly SOURCES: a file added under csrc/ but not
    listed would be silently left out of the library (or of the staleness check)."""
    import os

    from dualip_amd import _build

    on_disk = sorted(os.listdir(_build.CSRC))
    hips = [f for f in on_disk if f.endswith(".hip")]
    headers = [f for f in on_disk if f.endswith(".h")]
    assert sorted(_build.SOURCES) == hips
    listed_headers = sorted(os.path.basename(h) for h in _build.HEADERS if not h.startswith(".."))
    assert listed_headers == headers
    # the four translation units of the 256-wide fused kernel are one header compiled with two switches
    for name, lanes, f64 in (("matching_kernels4.hip", 0, 0), ("matching_kernels4_f64.hip", 0, 1), ("matching_kernels4_lanes.hip", 1, 0), ("matching_kernels4_lanes_f64.hip", 1, 1)):
        text = open(os.path.join(_build.CSRC, name)).read()
        assert f"#define DL_FUSED4_LANES {lanes}" in text and f"#define DL_FUSED4_F64 {f64}" in text and '#include "fused4_kernel.h"' in text


def test_local_shard_says_when_balanced_cannot_be_honoured_and_builds_empty_shards():
    """run_solver._local_shard: partition='balanced' on a map without contiguous range blocks falls back to the reference's cut WITH a
    warning (not silently); a rank whose share of every block is empty gets an empty shard instead of a torch.cat error."""
    import warnings

    import numpy as np
    import torch

    from dualip_amd.objectives.matching import MatchingInputArgs
    from dualip_amd.projections import create_projection_map
    from dualip_amd.run_solver import _local_shard

    n, m = 6, 4
    colptr = torch.arange(0, 2 * n + 1, 2)
    rows = torch.tensor([0, 1] * n)
    vals = torch.arange(1.0, 2 * n + 1)
    A = torch.sparse_csc_tensor(colptr, rows, vals, size=(m, n))
    C = torch.sparse_csc_tensor(colptr, rows, -vals, size=(m, n))
    # two range blocks of 3 columns each, 4 ranks: rank 3's share of both blocks is empty
    pm = {**create_projection_map("box", {"lower": 0.0, "upper": 1.0}, None, indices=range(0, 3)), **create_projection_map("simplex", {"z": 1.0}, None, indices=range(3, 6))}
    inp = MatchingInputArgs(A=A, c=C, projection_map=pm, b_vec=torch.ones(m))
    widths = []
    for r in range(4):
        sh = _local_shard(inp, r, 4, "cpu", "balanced")
        widths.append(int(sh.A.size(1)))
        assert sh.A.values().numel() == 2 * widths[-1] and sh.A.ccol_indices().numel() == widths[-1] + 1
    assert widths == [2, 2, 2, 0], widths
    # index LISTS: no blocks to share out -> the reference's contiguous cut, said out loud
    pm_list = create_projection_map("simplex", {"z": 1.0}, n, indices=[0, 1, 2, 3, 4, 5])
    inp2 = MatchingInputArgs(A=A, c=C, projection_map=pm_list, b_vec=torch.ones(m))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sh = _local_shard(inp2, 1, 4, "cpu", "balanced")
    assert any("partition='balanced'" in str(x.message) for x in w)
    assert int(sh.A.size(1)) == 2  # n // W (+1 for the first n % W ranks): 2, 2, 1, 1
    assert np.array_equal(sh.A.values().numpy(), vals.numpy()[4:8])


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_bench_partitions_cover_every_entity_once_for_any_world(world):
    """bench.py's three splits (the reference's n // W (+1) cut, the cost-weighted contiguous cut, the balanced per-block shares) for
    the world sizes of the reference's sweep (benchmark/run_scaling_benchmark.py:33-55: 1..4, three included) and the node's eight:
    every entity lands on exactly one rank and keeps its operator."""
    import bench
    from benchmark.synthetic import CHUNK_COLS

    n = 25 * CHUNK_COLS + 12345
    blocks = bench.projection_blocks("mixed", n, CHUNK_COLS)
    for partition in ("reference", "contiguous", "balanced"):
        seen = np.zeros(n, dtype=np.int32)
        for rank in range(world):
            ranges, pm = bench.shard_plan("mixed", n, world, rank, CHUNK_COLS, partition)
            local = 0
            for lo, hi in ranges:
                seen[lo:hi] += 1
                local += hi - lo
            assert sum(len(e.indices) for e in pm.values()) == local
            # the operator of a local column is the operator of the global column it came from
            pos = 0
            for lo, hi in ranges:
                kinds = {ptype for ptype, _, blo, bhi in blocks if blo < hi and bhi > lo}
                for e in pm.values():
                    if e.indices.start <= pos < e.indices.stop:
                        assert e.proj_type in kinds
                pos += hi - lo
        assert (seen == 1).all(), (partition, world, int((seen != 1).sum()))
