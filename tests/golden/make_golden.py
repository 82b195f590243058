#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, never on the GPU box).

Imports the reference implementation from /root/reference (read-only) with an empty
``mlflow`` stub on sys.path (the reference hard-imports mlflow, src/dualip/utils/mlflow_utils.py:5),
runs the reference's own matching objective / AGD maximizer / projections on small seeded problems
and writes the inputs + outputs as ``.npz`` fixtures next to this script.

The fixtures are data only (inputs and expected outputs).  Nothing from the reference's source
is stored.  Re-run with:  python tests/golden/make_golden.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _setup_reference_imports():
    stub = tempfile.mkdtemp(prefix="mlflow_stub_")
    os.makedirs(os.path.join(stub, "mlflow"), exist_ok=True)
    with open(os.path.join(stub, "mlflow", "__init__.py"), "w") as f:
        f.write("")
    sys.path.insert(0, stub)
    sys.path.insert(0, os.path.join(REF, "src"))
    sys.path.insert(0, os.path.join(REF, "benchmark"))
    return stub


_STUB = _setup_reference_imports()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402
from dualip.objectives.matching import (  # noqa: E402
    MatchingInputArgs,
    MatchingSolverDualObjectiveFunction,
    MatchingSolverDualObjectiveFunctionDistributed,
)
from dualip.optimizers.agd import AcceleratedGradientDescent  # noqa: E402
from dualip.preprocessing.precondition import jacobi_precondition  # noqa: E402
from dualip.projections.base import ProjectionEntry, create_projection_map, project  # noqa: E402
from dualip.utils.dist_utils import global_to_local_projection_map, split_tensors_to_devices  # noqa: E402
from generate_synthetic_data import generate_synthetic_matching_input_args  # noqa: E402

torch.set_num_threads(4)

DT = {"f32": torch.float32, "f64": torch.float64}


# --------------------------------------------------------------------------------------
# problems
# --------------------------------------------------------------------------------------
def problem_synthetic(S, D, sp, seed):
    """Reference generator (benchmark/generate_synthetic_data.py:345-470). Values are fp32-representable."""
    with tempfile.TemporaryDirectory() as tmp:
        args = generate_synthetic_matching_input_args(S, D, sp, device="cpu", dtype=torch.float64, seed=seed, cache_dir=tmp)
    return dict(
        m=D,
        n=S,
        colptr=args.A.ccol_indices().numpy().astype(np.int64),
        rowidx=args.A.row_indices().numpy().astype(np.int64),
        a=args.A.values().numpy().astype(np.float64),
        c=args.c.values().numpy().astype(np.float64),
        b=args.b_vec.numpy().astype(np.float64),
    )


def problem_movielens_like(n_users, n_movies, seed):
    """Shape of examples/movielens_matching/movies_lens_matching.py:60-130: a == 1, c == -rating, heavy-tailed
    ratings-per-user (a few users rate most of the catalogue), uniform capacity b."""
    rng = np.random.default_rng(seed)
    deg = np.clip(np.round(rng.lognormal(3.0, 1.2, size=n_users)).astype(np.int64), 0, n_movies)
    deg[rng.integers(0, n_users, size=5)] = 0  # a few empty users
    pop = rng.lognormal(0.0, 1.0, size=n_movies)
    pop /= pop.sum()
    cols, rows = [], []
    for u in range(n_users):
        if deg[u] == 0:
            continue
        r = np.sort(rng.choice(n_movies, size=deg[u], replace=False, p=pop))
        rows.append(r)
        cols.append(np.full(deg[u], u))
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    counts = np.bincount(cols, minlength=n_users)
    colptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(counts, out=colptr[1:])
    rating = rng.integers(1, 11, size=rows.shape[0]).astype(np.float64) * 0.5
    return dict(
        m=n_movies,
        n=n_users,
        colptr=colptr,
        rowidx=rows.astype(np.int64),
        a=np.ones(rows.shape[0], dtype=np.float64),
        c=-rating,
        b=np.full(n_movies, 3.0, dtype=np.float64),
    )


def to_args(p, dtype, projection_map, equality_mask=None, with_b=True):
    A = torch.sparse_csc_tensor(
        torch.from_numpy(p["colptr"]), torch.from_numpy(p["rowidx"]), torch.from_numpy(p["a"].copy()).to(dtype), size=(p["m"], p["n"])
    )
    C = torch.sparse_csc_tensor(
        torch.from_numpy(p["colptr"]), torch.from_numpy(p["rowidx"]), torch.from_numpy(p["c"].copy()).to(dtype), size=(p["m"], p["n"])
    )
    b = torch.from_numpy(p["b"].copy()).to(dtype) if with_b else None
    return MatchingInputArgs(A=A, c=C, projection_map=projection_map, b_vec=b, equality_mask=equality_mask)


def fnum(v):
    return float(v.item()) if hasattr(v, "item") else float(v)


# --------------------------------------------------------------------------------------
# G1: single calculate() calls
# --------------------------------------------------------------------------------------
SINGLE_MAPS = {
    "box01": ("box", {"lower": 0.0, "upper": 1.0}),
    "box_l0.05_u0.4": ("box", {"lower": 0.05, "upper": 0.4}),
    "simplex1": ("simplex", {"z": 1.0}),
    "simplex2.5": ("simplex", {"z": 2.5}),
    "cone_lower0": ("cone", {"lower": 0.0}),
    "cone_upper0.3": ("cone", {"upper": 0.3}),
}


def run_calculate(p, dtype, proj_type, proj_params, gamma, lam, batching):
    pm = create_projection_map(proj_type, dict(proj_params), p["n"])
    args = to_args(p, dtype, pm)
    obj = MatchingSolverDualObjectiveFunction(args, gamma=gamma, batching=batching)
    res = obj.calculate(torch.from_numpy(lam).to(dtype), gamma=gamma, save_primal=True)
    return dict(
        grad=res.dual_gradient.numpy().copy(),
        x=res.primal_var.numpy().copy(),
        scal=np.array(
            [
                fnum(res.dual_objective),
                fnum(res.reg_penalty),
                fnum(res.primal_objective),
                fnum(res.dual_val_times_grad),
                fnum(res.max_pos_slack),
                fnum(res.sum_pos_slack),
            ],
            dtype=np.float64,
        ),
    )


def lam_cases(p, seed):
    rng = np.random.default_rng(seed)
    m = p["m"]
    # fp32-representable so the same lambda feeds both precisions
    return {
        "zero": np.zeros(m),
        "small": rng.uniform(0, 0.01, m).astype(np.float32).astype(np.float64),
        "large": rng.uniform(0, 0.6, m).astype(np.float32).astype(np.float64),
    }


def make_g1(name, p, gammas, maps, seed):
    out = {k: v for k, v in p.items() if isinstance(v, np.ndarray)}
    out["m"] = np.int64(p["m"])
    out["n"] = np.int64(p["n"])
    lams = lam_cases(p, seed)
    for ln, lv in lams.items():
        out[f"lam_{ln}"] = lv
    cases = []
    for mk in maps:
        pt, pp = SINGLE_MAPS[mk]
        for g in gammas:
            for ln, lv in lams.items():
                for dn, dt in DT.items():
                    r = run_calculate(p, dt, pt, pp, g, lv, batching=False)
                    # batching on/off must agree for these operators (padding-independent); keep the check honest
                    rb = run_calculate(p, dt, pt, pp, g, lv, batching=True)
                    tol = 1e-9 if dn == "f64" else 2e-4
                    assert np.allclose(r["x"], rb["x"], rtol=tol, atol=tol), (mk, g, ln, dn)
                    key = f"{mk}|{g}|{ln}|{dn}"
                    cases.append(key)
                    for kk, vv in r.items():
                        out[f"{key}|{kk}"] = vv
    out["cases"] = np.array(cases)
    path = os.path.join(HERE, f"g1_{name}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", len(cases), "cases")


# --------------------------------------------------------------------------------------
# G2: AGD traces
# --------------------------------------------------------------------------------------
def run_trace(p, dtype, pm, gamma, iters, init_step, max_step, decay=None, eq_mask=None, jacobi=False, lam0=None):
    args = to_args(p, dtype, pm, equality_mask=None if eq_mask is None else torch.from_numpy(eq_mask))
    row_norms = None
    if jacobi:
        row_norms = jacobi_precondition(args.A, args.b_vec)
    obj = MatchingSolverDualObjectiveFunction(args, gamma=gamma, batching=False)
    solver = AcceleratedGradientDescent(
        max_iter=iters,
        gamma=gamma,
        initial_step_size=init_step,
        max_step_size=max_step,
        gamma_decay_type="step" if decay else None,
        gamma_decay_params=decay,
        save_primal=True,
        iteration_callback=lambda i, r: None,
    )
    lam = torch.zeros(p["m"], dtype=dtype) if lam0 is None else torch.from_numpy(lam0).to(dtype)
    res = solver.maximize(obj, lam)
    o = res.objective_result
    out = dict(
        dual_obj_log=np.array(res.dual_objective_log, dtype=np.float64),
        step_log=np.array(res.step_size_log, dtype=np.float64),
        dual_val=res.dual_val.numpy().copy(),
        grad=o.dual_gradient.numpy().copy(),
        x=o.primal_var.numpy().copy(),
        scal=np.array(
            [
                fnum(o.dual_objective),
                fnum(o.reg_penalty),
                fnum(o.primal_objective),
                fnum(o.dual_val_times_grad),
                fnum(o.max_pos_slack),
                fnum(o.sum_pos_slack),
            ]
        ),
        final_gamma=np.float64(solver.gamma),
    )
    if row_norms is not None:
        out["row_norms"] = row_norms.numpy().copy()
        out["A_scaled"] = args.A.values().numpy().copy()
        out["b_scaled"] = args.b_vec.numpy().copy()
    return out


def make_g2(name, p, seed):
    rng = np.random.default_rng(seed)
    out = {k: v for k, v in p.items() if isinstance(v, np.ndarray)}
    out["m"] = np.int64(p["m"])
    out["n"] = np.int64(p["n"])
    eq_mask = rng.uniform(size=p["m"]) < 0.2
    out["eq_mask"] = eq_mask
    n = p["n"]
    variants = {
        # name: (proj, params, gamma, iters, init_step, max_step, decay, eq, jacobi)
        "simplex1": ("simplex", {"z": 1.0}, 0.02, 60, 1e-3, 1e-1, None, False, False),
        "box01": ("box", {"lower": 0.0, "upper": 1.0}, 0.02, 60, 1e-3, 1e-1, None, False, False),
        "simplex1_decay": ("simplex", {"z": 1.0}, 0.08, 60, 1e-3, 1e-1, {"decay_steps": 10, "decay_factor": 0.5}, False, False),
        "simplex1_eq": ("simplex", {"z": 1.0}, 0.02, 60, 1e-3, 1e-1, None, True, False),
        "simplex1_jacobi": ("simplex", {"z": 1.0}, 0.02, 60, 1e-3, 1e-1, None, False, True),
        "simplex1_smallgamma": ("simplex", {"z": 1.0}, 1e-3, 40, 1e-3, 1e-1, None, False, False),
    }
    names = []
    for vn, (pt, pp, g, it, s0, s1, decay, eq, jac) in variants.items():
        for dn, dt in DT.items():
            pm = create_projection_map(pt, dict(pp), n)
            r = run_trace(p, dt, pm, g, it, s0, s1, decay=decay, eq_mask=eq_mask if eq else None, jacobi=jac)
            key = f"{vn}|{dn}"
            names.append(key)
            for kk, vv in r.items():
                out[f"{key}|{kk}"] = vv
            out[f"{key}|params"] = np.array(
                [g, it, s0, s1, decay["decay_steps"] if decay else 0, decay["decay_factor"] if decay else 0, float(eq), float(jac)]
            )
            out[f"{key}|proj"] = np.array([pt, *[f"{k}={v}" for k, v in pp.items()]])
    out["variants"] = np.array(names)
    path = os.path.join(HERE, f"g2_{name}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


# --------------------------------------------------------------------------------------
# G3: distributed (gloo / CPU) traces; G3m: mixed map via key-boundary split
# --------------------------------------------------------------------------------------
def _dist_worker(rank, world, port, payload, retq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    p, dn, shards, gamma, iters, s0, s1, lam_single = payload
    dtype = DT[dn]
    lo, hi, pm_local = shards[rank]
    sub = dict(
        m=p["m"],
        n=hi - lo,
        colptr=p["colptr"][lo : hi + 1] - p["colptr"][lo],
        rowidx=p["rowidx"][p["colptr"][lo] : p["colptr"][hi]],
        a=p["a"][p["colptr"][lo] : p["colptr"][hi]],
        c=p["c"][p["colptr"][lo] : p["colptr"][hi]],
        b=p["b"],
    )
    local = to_args(sub, dtype, pm_local, with_b=False)
    f = MatchingSolverDualObjectiveFunctionDistributed(
        local_matching_input_args=local, b_vec=torch.from_numpy(p["b"]).to(dtype), gamma=gamma, host_device="cpu", batching=False
    )
    result = {}
    if lam_single is not None:
        r = f.calculate(torch.from_numpy(lam_single).to(dtype), gamma=gamma, rank=rank)
        if rank == 0:
            result["single_grad"] = r.dual_gradient.numpy().copy()
            result["single_scal"] = np.array(
                [fnum(r.dual_objective), fnum(r.reg_penalty), 0.0, fnum(r.dual_val_times_grad), fnum(r.max_pos_slack), fnum(r.sum_pos_slack)]
            )
        # local primal slice (the reference never returns it in distributed mode; read the local scratch)
        result_x = f.local_objective.intermediate.values().numpy().copy()
        gathered = [None] * world
        dist.all_gather_object(gathered, result_x)
        if rank == 0:
            result["single_x"] = np.concatenate(gathered)
    if iters > 0:
        solver = AcceleratedGradientDescent(
            max_iter=iters, gamma=gamma, initial_step_size=s0, max_step_size=s1, iteration_callback=lambda i, r: None
        )
        res = solver.maximize(f, torch.zeros(p["m"], dtype=dtype), rank=rank)
        if rank == 0:
            result["dual_obj_log"] = np.array(res.dual_objective_log)
            result["step_log"] = np.array(res.step_size_log)
            result["dual_val"] = res.dual_val.numpy().copy()
            result["grad"] = res.objective_result.dual_gradient.numpy().copy()
    if rank == 0:
        retq.put(result)
    dist.barrier()
    dist.destroy_process_group()


_PORT = [29610]


def run_dist(p, dn, shards, gamma, iters, s0, s1, lam_single=None):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    _PORT[0] += 1
    world = len(shards)
    procs = [ctx.Process(target=_dist_worker, args=(r, world, _PORT[0], (p, dn, shards, gamma, iters, s0, s1, lam_single), q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = q.get()
    for pr in procs:
        pr.join()
    return res


def even_shards(n, world, proj_type, proj_params):
    """Contiguous split n//W (+1 for the first n%W), as src/dualip/utils/dist_utils.py:53-57."""
    sizes = [n // world + (1 if i < n % world else 0) for i in range(world)]
    shards, lo = [], 0
    for s in sizes:
        pm = create_projection_map(proj_type, dict(proj_params), s)
        shards.append((lo, lo + s, pm))
        lo += s
    return shards


def make_g3(name, p, seed):
    out = {k: v for k, v in p.items() if isinstance(v, np.ndarray)}
    out["m"] = np.int64(p["m"])
    out["n"] = np.int64(p["n"])
    n = p["n"]
    lam = lam_cases(p, seed)["large"]
    out["lam"] = lam
    keys = []
    gamma, iters, s0, s1 = 0.02, 40, 1e-3, 1e-1
    out["params"] = np.array([gamma, iters, s0, s1])
    for world in (2, 4, 8):  # 8 = the target machine's world size (benchmark/run_matching_benchmark_dist.py:33-193)
        for dn in DT:
            r = run_dist(p, dn, even_shards(n, world, "simplex", {"z": 1.0}), gamma, iters, s0, s1, lam_single=lam)
            key = f"simplex1|w{world}|{dn}"
            keys.append(key)
            for kk, vv in r.items():
                out[f"{key}|{kk}"] = vv
    # G3m: mixed map, first half box[0,1], second half simplex z=1 -- split AT the key boundary so every rank
    # holds a single-key map (the reference's single-process multi-key path is defective).
    half = n // 2
    out["mixed_boundary"] = np.int64(half)
    for dn in DT:
        shards = [
            (0, half, create_projection_map("box", {"lower": 0.0, "upper": 1.0}, half)),
            (half, n, create_projection_map("simplex", {"z": 1.0}, n - half)),
        ]
        r = run_dist(p, dn, shards, gamma, iters, s0, s1, lam_single=lam)
        key = f"mixed|w2|{dn}"
        keys.append(key)
        for kk, vv in r.items():
            out[f"{key}|{kk}"] = vv
    # the same mixed map on 8 ranks: four ranks share the box half, four the simplex half (every rank still single-key)
    for dn in DT:
        shards = []
        for lo0, hi0, pt, pp in ((0, half, "box", {"lower": 0.0, "upper": 1.0}), (half, n, "simplex", {"z": 1.0})):
            shards += [(lo0 + lo, lo0 + hi, pm) for lo, hi, pm in even_shards(hi0 - lo0, 4, pt, pp)]
        r = run_dist(p, dn, shards, gamma, iters, s0, s1, lam_single=lam)
        key = f"mixed|w8|{dn}"
        keys.append(key)
        for kk, vv in r.items():
            out[f"{key}|{kk}"] = vv
    out["keys"] = np.array(keys)
    path = os.path.join(HERE, f"g3_{name}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


# --------------------------------------------------------------------------------------
# G4: beta sequence; GP: projection operators on dense blocks; GS: 5x5 Scala fixture trace
# --------------------------------------------------------------------------------------
def make_g4():
    s = AcceleratedGradientDescent(max_iter=10000, gamma=1e-3)
    path = os.path.join(HERE, "g4_beta_seq.npz")
    np.savez_compressed(path, beta=s.beta_seq.numpy())
    print("wrote", path)


def make_gp(seed):
    rng = np.random.default_rng(seed)
    out = {}
    blocks = {
        "spread": rng.normal(0.0, 3.0, size=(12, 200)),
        "tight": rng.uniform(0.0, 0.3, size=(9, 150)),
        "neg": rng.normal(-0.5, 0.4, size=(7, 100)),
        "single_row": rng.normal(0.5, 1.0, size=(1, 50)),
        "ties": np.round(rng.uniform(0, 1, size=(8, 120)) * 4) / 4,
        "long": rng.normal(0.0, 0.05, size=(150, 40)),
    }
    ops = {
        "simplex_z1": ("simplex", {"z": 1.0}),
        "simplex_z0.3": ("simplex", {"z": 0.3}),
        "simplex_eq_z1": ("simplex_eq", {"z": 1.0}),
        "simplex_bisect_z1": ("simplex", {"z": 1.0, "method": "bisection_search"}),
        "box": ("box", {"lower": -0.2, "upper": 0.7}),
        "cone_lo": ("cone", {"lower": 0.1}),
        "cone_up": ("cone", {"upper": 0.1}),
    }
    for bn, bv in blocks.items():
        bv = bv.astype(np.float32).astype(np.float64)
        out[f"in|{bn}"] = bv
        for on, (pt, pp) in ops.items():
            for dn, dt in DT.items():
                y = project(pt, **pp)(torch.from_numpy(bv).to(dt))
                out[f"out|{bn}|{on}|{dn}"] = y.numpy().copy()
    out["blocks"] = np.array(list(blocks))
    out["ops"] = np.array(list(ops))
    path = os.path.join(HERE, "gp_projections.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def make_g7(seed):
    """MovieLens-shaped stand-in: a == 1, c = -rating, simplex z=1, gamma = 0.1."""
    p = problem_movielens_like(3000, 400, seed)
    out = {k: v for k, v in p.items() if isinstance(v, np.ndarray)}
    out["m"] = np.int64(p["m"])
    out["n"] = np.int64(p["n"])
    pm = create_projection_map("simplex", {"z": 1.0}, p["n"])
    keys = []
    for dn, dt in DT.items():
        r = run_trace(p, dt, pm, 0.1, 120, 1e-4, 1e-2)
        for kk, vv in r.items():
            out[f"{dn}|{kk}"] = vv
        keys.append(dn)
    out["params"] = np.array([0.1, 120, 1e-4, 1e-2])
    path = os.path.join(HERE, "g7_movielens_like.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", "nnz", p["rowidx"].shape[0], "max col", np.diff(p["colptr"]).max())


def main():
    torch.manual_seed(0)
    syn = problem_synthetic(2000, 50, 0.1, 42)
    if "--only-g3" in sys.argv:  # (every fixture is a pure function of its seeds: regenerating one leaves the others as they are)
        make_g3("syn2000", syn, seed=10)
        return
    print("syn2000 nnz", syn["rowidx"].shape[0], "max col nnz", np.diff(syn["colptr"]).max(), "empty", (np.diff(syn["colptr"]) == 0).sum())
    make_g1("syn2000", syn, gammas=[1e-3, 0.02, 0.1], maps=list(SINGLE_MAPS), seed=7)
    longp = problem_synthetic(150, 400, 0.35, 43)
    print("long nnz", longp["rowidx"].shape[0], "max col nnz", np.diff(longp["colptr"]).max())
    make_g1("long", longp, gammas=[0.02, 0.5], maps=["simplex1", "simplex2.5", "box01"], seed=8)
    make_g2("syn2000", syn, seed=9)
    make_g3("syn2000", syn, seed=10)
    make_g4()
    make_gp(seed=11)
    make_g7(seed=12)


if __name__ == "__main__":
    main()
