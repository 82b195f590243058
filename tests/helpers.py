"""Shared fixture helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> (proj_type, params) used by tests/golden/make_golden.py
SINGLE_MAPS = {
    "box01": ("box", {"lower": 0.0, "upper": 1.0}),
    "box_l0.05_u0.4": ("box", {"lower": 0.05, "upper": 0.4}),
    "simplex1": ("simplex", {"z": 1.0}),
    "simplex2.5": ("simplex", {"z": 2.5}),
    "cone_lower0": ("cone", {"lower": 0.0}),
    "cone_upper0.3": ("cone", {"upper": 0.3}),
}
NP_DT = {"f32": np.float32, "f64": np.float64}
# parity tolerances (relative to the magnitude of the compared vector)
RTOL = {"f32": 2e-4, "f64": 1e-9}


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def problem(z):
    return dict(m=int(z["m"]), n=int(z["n"]), colptr=z["colptr"], rowidx=z["rowidx"], a=z["a"], c=z["c"], b=z["b"])


def relerr(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    scale = max(1.0, float(np.max(np.abs(want)))) if want.size else 1.0
    return float(np.max(np.abs(got - want))) / scale if want.size else 0.0


def scala_5x5():
    """The 5x5 known-answer problem of the reference's tests/objectives/test_dualip_matching_simplex.py:10-99
    (values are data of the test; the matrix is its own transpose-by-construction layout: column j of the CSC
    matrix is row j of the dense "user x item" table)."""
    a = np.array(
        [
            [0.307766110869125, 0.483770735096186, 0.624996477039531, 0.669021712383255, 0.535811153938994],
            [0.257672501029447, 0.812402617651969, 0.882165518123657, 0.204612161964178, 0.710803845431656],
            [0.552322433330119, 0.370320537127554, 0.28035383997485, 0.357524853432551, 0.538348698290065],
            [0.0563831503968686, 0.546558595029637, 0.398487901547924, 0.359475114848465, 0.74897222686559],
            [0.468549283919856, 0.170262051047757, 0.76255108229816, 0.690290528349578, 0.420101450523362],
        ],
        dtype=np.float32,
    )
    # CSC of a.T : column j holds a[j, :] with rows 0..4
    colptr = np.arange(0, 26, 5, dtype=np.int64)
    rowidx = np.tile(np.arange(5, dtype=np.int64), 5)
    vals = a.reshape(-1)
    return dict(m=5, n=5, colptr=colptr, rowidx=rowidx, a=vals.copy(), c=-vals, b=np.full(5, 0.7, dtype=np.float32))


SCALA_GOLDEN = [(2, -3.6010155991401818), (16, -3.60842718733725), (23, -3.5080258013053136), (29, -3.4868496294227143)]


# ---------------------------------------------------------------------------------------------------------
# torch-side helpers
# ---------------------------------------------------------------------------------------------------------
def torch_args(p, dtype, projection_map, device, with_b=True, equality_mask=None):
    """MatchingInputArgs of the package under test from a numpy problem dict."""
    import torch

    from dualip_amd.objectives.matching import MatchingInputArgs

    td = {"f32": torch.float32, "f64": torch.float64}[dtype] if isinstance(dtype, str) else dtype
    colptr = torch.from_numpy(np.ascontiguousarray(p["colptr"], dtype=np.int64))
    rowidx = torch.from_numpy(np.ascontiguousarray(p["rowidx"], dtype=np.int64))
    A = torch.sparse_csc_tensor(colptr, rowidx, torch.from_numpy(np.array(p["a"], dtype=np.float64)).to(td), size=(p["m"], p["n"])).to(device)
    C = torch.sparse_csc_tensor(colptr, rowidx, torch.from_numpy(np.array(p["c"], dtype=np.float64)).to(td), size=(p["m"], p["n"])).to(device)
    b = torch.from_numpy(np.array(p["b"], dtype=np.float64)).to(td).to(device) if with_b else None
    em = None if equality_mask is None else torch.from_numpy(np.asarray(equality_mask)).to(device)
    return MatchingInputArgs(A=A, c=C, projection_map=projection_map, b_vec=b, equality_mask=em)


def sub_problem(p, lo, hi):
    k0, k1 = int(p["colptr"][lo]), int(p["colptr"][hi])
    return dict(m=p["m"], n=hi - lo, colptr=p["colptr"][lo : hi + 1] - k0, rowidx=p["rowidx"][k0:k1], a=p["a"][k0:k1], c=p["c"][k0:k1], b=p["b"])


class OracleLocalObjective:
    """CPU stand-in for the per-rank fused pass (tests only): same surface as the native local objective
    (``calculate_packed`` / ``finish``), arithmetic by oracle/.  Lets the world_size-2 gloo tests exercise the
    exchange + update logic of the distributed objective without a GPU."""

    def __init__(self, p, projs, gamma, np_dtype, col_proj=None):
        import torch

        self.p, self.projs, self.gamma, self.np_dtype, self.col_proj = p, projs, gamma, np_dtype, col_proj
        self.m = p["m"]
        self.device = torch.device("cpu")
        self.dtype = torch.float32 if np_dtype == np.float32 else torch.float64

    def calculate_packed(self, dual_val, gamma=None, x_out=None):
        import torch

        import oracle

        if gamma is not None:
            self.gamma = gamma
        ax, obj0, ssq, _ = oracle.matching_calculate(
            self.p["m"], self.p["n"], self.p["colptr"], self.p["rowidx"], self.p["a"], self.p["c"], dual_val.numpy(), self.gamma,
            self.projs, col_proj=self.col_proj, dtype=self.np_dtype, want_x=False,
        )
        return torch.from_numpy(np.concatenate([ax.astype(np.float64), [obj0, ssq]]))

    def finish(self, packed, dual_val, b_vec):
        import torch

        from dualip_amd.types import ObjectiveResult
        from oracle import agd_oracle

        pk = packed.numpy()
        grad, obj, reg, dvtg, mx, sm = agd_oracle.epilogue(pk[: self.m], pk[self.m], pk[self.m + 1], dual_val.numpy(), b_vec.numpy(), self.gamma, self.np_dtype)
        t = lambda v: torch.tensor(float(v), dtype=self.dtype)  # noqa: E731
        return ObjectiveResult(
            dual_gradient=torch.from_numpy(np.asarray(grad, dtype=self.np_dtype)), dual_objective=t(obj), reg_penalty=t(reg),
            dual_val_times_grad=t(dvtg), max_pos_slack=t(mx), sum_pos_slack=t(sm),
        )


def lp_small_entries(z):
    """The projection map of tests/golden/make_golden_lp.py:small_map as (proj_type, params, indices) triples."""
    return [
        ("box", {"lower": -0.5, "upper": 1.5}, z["idx_two"]),
        ("box", {}, z["idx_unit"]),
        ("cone", {"lower": 0.0}, z["idx_lo"]),
        ("cone", {"upper": 0.75}, z["idx_up"]),
        ("box", {"lower": 0.25, "upper": 2.0}, z["idx_lu_names"]),
    ]


def padded_eq_entries(p, zz, batching):
    """Oracle description of the reference's zero-padded ``simplex_eq`` blocks: one oracle entry per nnz-bucket
    (``batching``) or a single one, each with ``lblock`` = the longest column it holds (matching.py:87-114,
    sparse_utils.py:185-186).  Returns (entries, lblocks, col_proj)."""
    lens = np.diff(p["colptr"])
    if not batching:
        return [("simplex_eq", {"z": zz})], [int(lens.max())], np.zeros(p["n"], dtype=np.int32)
    th = [0]
    i = 1
    while 2**i <= p["m"]:
        th.append(2**i)
        i += 1
    th.append(p["m"] + 1)
    bucket = np.searchsorted(np.array(th), lens, side="left")  # torch.bucketize(right=False)
    ids = sorted(set(int(b) for b in bucket[lens > 0]))
    remap = {b: k for k, b in enumerate(ids)}
    col_proj = np.array([remap.get(int(b), 0) for b in bucket], dtype=np.int32)
    lblocks = [int(lens[bucket == b].max()) for b in ids]
    return [("simplex_eq", {"z": zz})] * len(ids), lblocks, col_proj


def verify_at_size(dtype_name, gamma, inp, pm_local, f, local, lam, rank=0, world=1, sharded=False, device="cuda:0", comm_backend=None, length_classes=None,
                   skip_route_check=False):
    """Correctness at the benchmark size (bench.py runs it outside every timed region -> aux.verified; tests/test_gpu_fullsize.py
    runs it on the 10M-entity configurations).  Returns {"ok": bool, "checks": [...]}.

    1. the oracle (oracle/, the CPU restatement pinned to the reference's goldens) on slabs of 5000 columns: one inside
       every projection block, one straddling every block boundary, and the last columns of the arrays (largest offsets);
    2. A x, c.x, sum x^2 recomputed from the returned primal with torch ops (float64, chunked);
    3. N = 1: the sharded route (this shard split into two kernel handles + the exchange) against the single objective;
       N > 1: this library's exchange against torch.distributed's all-reduce of the same local sums, and the duals of all
       ranks bit-identical.

    ``length_classes``: [(lo, hi), ...] -- for every class that has a column of lo <= length <= hi, one more oracle slab of 400
    columns around such a column (shapes whose kernel plan depends on the column length: each plan's columns get checked).
    The names of those checks carry ``length class [lo, hi]``."""
    import torch
    import torch.distributed as dist

    import oracle

    out = {"ok": True, "checks": []}
    m, gamma = local.m, float(gamma)
    npdt = np.float32 if dtype_name == "f32" else np.float64
    tol_x = 2e-4 if dtype_name == "f32" else 1e-9

    def note(name, err, tol):
        good = bool(err <= tol)
        out["checks"].append({"name": name, "err": float(err), "tol": tol, "ok": good})
        out["ok"] = out["ok"] and good

    A, C = inp.A, inp.c
    colptr, rows, a_vals, c_vals = A.ccol_indices(), A.row_indices(), A.values(), C.values()
    n_local = A.shape[1]
    packed = local.calculate_packed(lam, gamma, x_out=local._primal_buffer()).clone()
    x = local._primal_buffer()
    lam_h = lam.cpu().numpy()
    entries = list(pm_local.items())
    bounds = []
    for _, e in entries:
        idx = e.indices
        bounds.append((idx.start, idx.stop) if isinstance(idx, range) else (int(min(idx)), int(max(idx)) + 1))
    slabs = []
    W = 5000
    gsl = torch.Generator().manual_seed(7)
    for q, (lo, hi) in enumerate(bounds):
        if hi - lo > W:
            s0 = lo + int(torch.randint(0, hi - lo - W, (1,), generator=gsl))
            slabs.append((f"inside entry {q} ({entries[q][1].proj_type})", s0, s0 + W))
    for q in range(len(bounds) - 1):
        cut = bounds[q][1]
        if cut == bounds[q + 1][0] and cut - W // 2 >= 0 and cut + W // 2 <= n_local:
            slabs.append((f"straddling the cut between entries {q} and {q + 1}", cut - W // 2, cut + W // 2))
    if n_local > W:
        slabs.append(("last columns of the arrays", n_local - W, n_local))
    if length_classes:
        lens_d = colptr[1:] - colptr[:-1]
        for lo_len, hi_len in length_classes:
            cand = torch.nonzero((lens_d >= lo_len) & (lens_d <= hi_len)).flatten()
            if cand.numel() == 0:
                continue
            j = int(cand[int(torch.randint(0, cand.numel(), (1,), generator=gsl))])
            s0 = max(0, min(j - 200, n_local - 400))
            slabs.append((f"length class [{lo_len}, {hi_len}] (column {j}, {int(lens_d[j])} non-zeros)", s0, min(n_local, s0 + 400)))
    for name, lo, hi in slabs:
        cp = colptr[lo : hi + 1].cpu().numpy().astype(np.int64)
        k0, k1 = int(cp[0]), int(cp[-1])
        cproj = np.full(hi - lo, -1, dtype=np.int32)
        for q, (blo, bhi) in enumerate(bounds):
            a0, a1 = max(lo, blo), min(hi, bhi)
            if a1 > a0:
                cproj[a0 - lo : a1 - lo] = q
        projs = [(e.proj_type, e.proj_params) for _, e in entries]
        _, _, _, xo = oracle.matching_calculate(m, hi - lo, cp - k0, rows[k0:k1].cpu().numpy().astype(np.int64), a_vals[k0:k1].cpu().numpy(),
                                                c_vals[k0:k1].cpu().numpy(), lam_h, gamma, projs, col_proj=cproj, dtype=npdt)
        xs = x[k0:k1].cpu().numpy()
        scale = max(float(np.abs(xo).max()), 1e-30)
        note(f"oracle slab [{lo}, {hi}) {name}, non-zeros [{k0}, {k1})", float(np.abs(xs - xo).max()) / scale, tol_x)
    # 2. the sums, recomputed from the primal
    ax = torch.zeros(m, dtype=torch.float64, device=device)
    cx = torch.zeros((), dtype=torch.float64, device=device)
    xx = torch.zeros((), dtype=torch.float64, device=device)
    step = 1 << 26
    for k0 in range(0, x.numel(), step):
        xs = x[k0 : k0 + step].double()
        ax.index_add_(0, rows[k0 : k0 + step].long(), a_vals[k0 : k0 + step].double() * xs)
        cx += (c_vals[k0 : k0 + step].double() * xs).sum()
        xx += (xs * xs).sum()
    tol_s = 1e-5 if dtype_name == "f32" else 1e-11
    note("A x recomputed from the primal (torch, float64)", float((ax - packed[:m]).abs().max() / ax.abs().max().clamp_min(1e-30)), tol_s)
    note("c.x recomputed from the primal", float((cx - packed[m]).abs() / cx.abs().clamp_min(1e-30)), tol_s)
    note("sum x^2 recomputed from the primal", float((xx - packed[m + 1]).abs() / xx.abs().clamp_min(1e-30)), tol_s)
    # 3. sharded against single / this library's exchange against torch.distributed's
    if skip_route_check:
        pass
    elif not sharded:
        from dualip_amd.objectives.matching import MatchingInputArgs, MatchingSolverDualObjectiveFunctionDistributed

        # two blocks, each with its share of EVERY projection entry (the partition bench.py gives the ranks of an N > 1 run)
        blocks = []
        for part in range(2):
            pos, pmb, A_parts = 0, {}, []
            for q, (blo, bhi) in enumerate(bounds):
                mid = blo + (bhi - blo) // 2
                lo, hi = (blo, mid) if part == 0 else (mid, bhi)
                k0, k1 = int(colptr[lo]), int(colptr[hi])
                sub_ptr = (colptr[lo : hi + 1] - k0)
                A_parts.append((sub_ptr, rows[k0:k1], a_vals[k0:k1], c_vals[k0:k1], hi - lo))
                key, e = entries[q]
                pmb[key] = type(e)(proj_type=e.proj_type, proj_params=e.proj_params, indices=range(pos, pos + hi - lo))
                pos += hi - lo
            ptrs, off = [torch.zeros(1, dtype=colptr.dtype, device=device)], 0
            for sp, r_, a_, c_, w_ in A_parts:
                ptrs.append(sp[1:] + off)
                off += int(r_.numel())
            cp_b = torch.cat(ptrs)
            r_b = torch.cat([t[1] for t in A_parts])
            a_b = torch.cat([t[2] for t in A_parts])
            c_b = torch.cat([t[3] for t in A_parts])
            Ab = torch.sparse_csc_tensor(cp_b, r_b, a_b, size=(m, pos), check_invariants=False)
            Cb = torch.sparse_csc_tensor(cp_b, r_b, c_b, size=(m, pos), check_invariants=False)
            blocks.append(MatchingInputArgs(A=Ab, c=Cb, projection_map=pmb, b_vec=None))
        fd = MatchingSolverDualObjectiveFunctionDistributed(blocks, inp.b_vec, gamma, host_device=device, comm_backend=comm_backend)
        r_sh = fd.calculate(lam, gamma=gamma)
        r_1 = f.calculate(lam, gamma=gamma)
        g1 = r_1.dual_gradient.double()
        note("sharded route (two blocks + exchange, world 1) against the single objective: gradient",
             float((r_sh.dual_gradient.double() - g1).abs().max() / g1.abs().max().clamp_min(1e-30)), 1e-6 if dtype_name == "f32" else 1e-12)
        note("... dual objective", abs(float(r_sh.dual_objective) - float(r_1.dual_objective)) / max(abs(float(r_1.dual_objective)), 1e-30), 1e-6 if dtype_name == "f32" else 1e-12)
        out["sharded_backend"] = fd.communicator().backend if fd.communicator() is not None else "torch.distributed"
        del fd, blocks
    else:
        ours = f.calculate_packed(lam, gamma).clone()
        ref = packed.clone()
        for blk in getattr(f, "more_blocks", []):
            ref += blk.calculate_packed(lam, gamma)
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        note("this library's exchange against torch.distributed all_reduce of the same local sums",
             float((ours - ref).abs().max() / ref.abs().max().clamp_min(1e-30)), 1e-12)
        digest = lam.view(torch.int32 if lam.dtype == torch.float32 else torch.int64).to(torch.int64).sum().double()
        lo_, hi_ = digest.clone(), digest.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        note("duals identical on all ranks (byte checksum spread)", float(hi_ - lo_), 0.0)
    torch.cuda.synchronize()
    return out


def gather_results(procs, q, timeout=420):
    """One result per worker process from a multiprocessing SimpleQueue -- WITHOUT blocking forever when a worker dies before it reports
    (a bare ``q.get()`` does: the first world-8 run of this suite sat in one until the box's time limit).  Raises as soon as a worker has
    exited with a non-zero code, or after ``timeout`` seconds, and ends the surviving workers (they would otherwise wait for the dead one
    inside a collective)."""
    import time

    out, t0 = [], time.time()
    while len(out) < len(procs):
        if not q.empty():
            out.append(q.get())
            continue
        failed = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
        if failed or time.time() - t0 > timeout:
            for p in procs:
                if p.is_alive():
                    p.terminate()
            raise AssertionError(f"{len(out)} of {len(procs)} workers reported; exit codes {[p.exitcode for p in procs]}" + ("" if failed else f" (no report within {timeout} s)"))
        time.sleep(0.05)
    return out


def retry_once_if_stalled(test):
    """Decorator for the tests that run SEVERAL RANKS AS PROCESSES ON THE ONE GPU of the test box.  Sharing a device, the ranks are time-sliced
    against each other, and now and then an in-kernel wait of the exchange runs into its bound -- the exchange then reports a timed-out wait on
    every rank, as it should (profiles/r05_world8_on_one_gpu.md).  One process per GPU never waits on a time-sliced peer.  A failure whose
    text -- the exception, or what the worker processes wrote to stderr -- says ``timed out`` is therefore reported as a WARNING carrying the
    first failure, and the test body runs once more; any other failure (a wrong number, a wrong plan, a second stall) is raised as it is."""
    import functools
    import inspect
    import sys
    import warnings

    @functools.wraps(test)
    def wrapper(*args, capfd, **kwargs):
        try:
            return test(*args, **kwargs)
        except (AssertionError, RuntimeError) as exc:
            err = capfd.readouterr().err
            sys.stderr.write(err)  # (what was captured so far stays visible in the report)
            if "timed out" not in f"{exc}\n{err}":
                raise
            warnings.warn(f"{test.__name__}: an in-kernel wait of the exchange timed out with the ranks time-sliced on one device; first failure: {str(exc)[:300]} -- running the test once more")
        return test(*args, **kwargs)

    sig = inspect.signature(test)
    wrapper.__signature__ = sig.replace(parameters=list(sig.parameters.values()) + [inspect.Parameter("capfd", inspect.Parameter.KEYWORD_ONLY)])
    return wrapper
