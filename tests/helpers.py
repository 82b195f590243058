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


from benchmark.verify import verify_at_size  # noqa: E402,F401  (moved out of the test package: bench.py uses it too)


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


# the exchange's own wording for an expired bounded wait (csrc/comm.hip: dl_comm_check / utils/comm.py: _STATUS[TIMED_OUT])
EXCHANGE_STALL_MESSAGES = ("a P2P exchange timed out waiting for another rank's partial sums", "a wait for another rank's partial sums timed out")
RETRIES = []           # names of the tests that were re-run after such a stall in this session
MAX_STALL_RETRIES = 2  # more than this many in one session fails it (tests/conftest.py: pytest_sessionfinish)


def retry_once_if_stalled(test):
    """Decorator for the tests that run SEVERAL RANKS AS PROCESSES ON THE ONE GPU of the test box.  Sharing a device, the ranks are time-sliced
    against each other, and now and then an in-kernel wait of the exchange runs into its bound -- the exchange then reports a timed-out wait on
    every rank, as it should (profiles/r05_world8_on_one_gpu.md).  One process per GPU never waits on a time-sliced peer.  A failure whose
    text -- the exception, or what the worker processes wrote to stderr -- carries one of the EXCHANGE'S OWN two messages for a bounded wait
    that expired (csrc/comm.hip: dl_comm_check; utils/comm.py: _STATUS -- not any "timed out": a gloo, subprocess or pytest timeout is a
    failure) is therefore reported as a WARNING carrying the first failure, and the test body runs once more; any other failure (a wrong
    number, a wrong plan, a second stall) is raised as it is.  Retries are counted per session (``RETRIES``); tests/conftest.py fails the
    session when more than ``MAX_STALL_RETRIES`` tests needed one -- a stall that common is a defect, not time-slicing."""
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
            if not any(msg in f"{exc}\n{err}" for msg in EXCHANGE_STALL_MESSAGES):
                raise
            RETRIES.append(test.__name__)
            warnings.warn(f"{test.__name__}: an in-kernel wait of the exchange timed out with the ranks time-sliced on one device; first failure: {str(exc)[:300]} -- running the test once more")
        return test(*args, **kwargs)

    sig = inspect.signature(test)
    wrapper.__signature__ = sig.replace(parameters=list(sig.parameters.values()) + [inspect.Parameter("capfd", inspect.Parameter.KEYWORD_ONLY)])
    return wrapper
