"""Generic-LP ("miplib2017") dual objective on the device.

Reference: src/dualip/objectives/miplib.py -- MIPLIBInputArgs :11-25, MIPLIB2017ObjectiveFunction :28-230.

    z = -1/gamma (A^T lambda' + c)          lambda' = lambda / row_norms when Jacobi-preconditioned (miplib.py:74-77)
    x = projection_map applied to z         (point-wise bounds for every shipped driver: read_mps_data.py:173-188)
    dual_gradient  = [1/row_norms] (A x - b)
    reg_penalty    = gamma/2 ||x||^2,   dual_objective = c.x + reg + lambda'.(A x - b)

x has one entry per variable.  ``A`` may be dense, COO, CSR or CSC (the reference accepts dense and COO); it is converted
once to CSC (for A^T lambda) and CSR (for A x) with int32 indices and handed to ``dl_lp_*`` (include/dualip_hip.h), which
borrows the device arrays.  Differences from the reference, on purpose:
  * bounds are read from ``lower``/``upper`` AND from ``l``/``u`` (the reference reads ``l``/``u`` for the convergence
    bound, miplib.py:117-120, while its MPS reader writes ``lower``/``upper``, read_mps_data.py:183-187); NaN = absent;
  * Jacobi preconditioning also works for sparse A (the reference raises, miplib.py:50-51);
  * ``invert_jacobi_precondition`` exists (run_solver.py:141 calls it; the reference class does not define it).
Maps whose entries are not point-wise bounds (a registered simplex over an index set, a user operator) take the two-call
route: ``dl_lp_primal`` (z) -> the operators on their index sets (as miplib.py:80-92) -> ``dl_lp_gradient``.
"""
import ctypes
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from dualip_amd import _hip
from dualip_amd.objectives.base import BaseInputArgs, BaseObjective
from dualip_amd.projections.base import ProjectionEntry, project
from dualip_amd.types import ObjectiveResult


@dataclass
class MIPLIBInputArgs(BaseInputArgs):
    """Input bundle of the generic-LP objective (reference miplib.py:11-25)."""

    A: torch.Tensor
    c: torch.Tensor
    projection_map: Dict[str, ProjectionEntry]
    b_vec: torch.Tensor
    equality_mask: Optional[torch.Tensor]

    def __post_init__(self):
        super().__post_init__()


def _param(params: dict, *names):
    for k in names:
        if k in params and params[k] is not None:
            v = float(params[k])
            if v == v:  # NaN = bound absent
                return v
    return None


def _box_bounds(params: dict):
    """(lower, upper) of a ``box`` entry as floats, -inf / +inf where a bound is absent.  A bound whose key is MISSING takes the
    operator's default -- ``BoxProjection(lower=0.0, upper=1.0)``, box.py:12-13: ``{"upper": 1}`` clamps to [0, 1] in the reference
    (tests/test_equality_constraints.py:39) -- unless the entry uses the ``l`` / ``u`` spelling of the reference's bound reader
    (miplib.py:111-121), where a missing key means "no bound"; a key that is present with NaN / None is an absent bound either way."""
    short = "l" in params or "u" in params
    out = []
    for long_name, short_name, default, absent in (("lower", "l", 0.0, float("-inf")), ("upper", "u", 1.0, float("inf"))):
        if long_name in params or short_name in params:
            v = _param(params, long_name, short_name)
            out.append(absent if v is None else v)
        else:
            out.append(absent if short else default)
    return out[0], out[1]


def _as_index_tensor(indices, device) -> torch.Tensor:
    if isinstance(indices, torch.Tensor):
        return indices.to(device=device, dtype=torch.long)
    return torch.as_tensor(list(indices) if isinstance(indices, range) else indices, dtype=torch.long, device=device)


def _sparse_forms(A: torch.Tensor):
    """(colptr, rowidx, vals_csc, rowptr, colidx, vals_csr) on the CPU, explicit zeros dropped for dense input."""
    A_cpu = A.detach().cpu()
    if A_cpu.layout == torch.sparse_coo:
        A_cpu = A_cpu.coalesce()
    csr = A_cpu.to_sparse_csr() if A_cpu.layout != torch.sparse_csr else A_cpu
    csc = A_cpu.to_sparse_csc() if A_cpu.layout != torch.sparse_csc else A_cpu
    return (
        csc.ccol_indices().to(torch.int64).contiguous(),
        csc.row_indices().to(torch.int32).contiguous(),
        csc.values().contiguous(),
        csr.crow_indices().to(torch.int64).contiguous(),
        csr.col_indices().to(torch.int32).contiguous(),
        csr.values().contiguous(),
    )


class MIPLIB2017ObjectiveFunction(BaseObjective):
    """Dual gradient, objective and regularisation penalty of a general LP with per-variable projections."""

    _dualip_native = True   # the maximizer keeps x, y, history and logs on the device ...
    _dualip_packed = True   # ... and drives this objective through calculate_packed_ptr (no sharding, no collective)

    def __init__(self, miplib_input_args: MIPLIBInputArgs, use_jacobi_precondition: bool = False):
        args = miplib_input_args
        self.A = args.A
        # CPU-resident inputs (solve_miplib_dataset.py:58 runs with host_device="cpu"): c and b are copied to the current ROCm device (A's
        # CSC / CSR forms are built on the host and moved there in any case); calculate() answers on the device of the duals it is given
        dev = args.c.device if args.c.is_cuda else (args.b_vec.device if args.b_vec.is_cuda else _hip.compute_device())
        self.c = _hip.stage(args.c, "c", dev)
        self.b_vec = _hip.stage(args.b_vec, "b_vec", dev)
        self.projection_map = args.projection_map
        self.equality_mask = args.equality_mask
        self.use_jacobi_precondition = bool(use_jacobi_precondition)
        self.device = self.c.device
        self.dtype = self.c.dtype
        if self.dtype not in (torch.float32, torch.float64):
            raise ValueError("c must be float32 or float64")
        if self.A.dim() != 2:
            raise ValueError("A must be a matrix")
        self.m, self.n = int(self.A.shape[0]), int(self.A.shape[1])
        if self.c.shape != (self.n,) or self.b_vec.shape != (self.m,):
            raise ValueError("c must have one entry per column of A and b_vec one per row")
        forms = _sparse_forms(self.A)
        if forms[2].dtype != self.dtype or self.b_vec.dtype != self.dtype:
            raise ValueError("A, c and b_vec must have the same dtype")
        (self._colptr, self._rowidx, self._vals_csc, self._rowptr, self._colidx, self._vals_csr) = (t.to(self.device) for t in forms)
        self.nnz = int(self._vals_csc.numel())
        self._c_dev = self.c.contiguous()
        self.lower, self.upper = self._construct_variable_lower_upper_bound()
        self._lo, self._hi, self._generic_entries = self._bounds_for_kernel()

        if self.use_jacobi_precondition:
            rows = torch.repeat_interleave(torch.arange(self.m, device=self.device), self._rowptr[1:] - self._rowptr[:-1])
            sq = torch.zeros(self.m, dtype=self.dtype, device=self.device).index_add_(0, rows, self._vals_csr * self._vals_csr)
            norms = torch.sqrt(sq)
            self.row_norms = torch.where(norms == 0, torch.ones_like(norms), norms)  # all-zero rows are left alone (miplib.py:54-55)
            self._inv_norm = (1 / self.row_norms).contiguous()
            self.b_step = (self._inv_norm * self.b_vec).contiguous()
        else:
            self.row_norms = None
            self._inv_norm = None
            self.b_step = self.b_vec.contiguous()

        self._lib = _hip.load()
        self._handle = self._create(self._inv_norm)
        self._handle_raw = self._create(None) if self._inv_norm is not None else self._handle
        self._packed = torch.zeros(self.m + 2, dtype=torch.float64, device=self.device)
        self._scal = torch.zeros(6, dtype=torch.float64, device=self.device)
        self._primal = None

    # ------------------------------------------------------------------------------------------------------
    def _create(self, inv_norm):
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = self._lib.dl_lp_create(
                ctypes.byref(handle), self.m, self.n, self.nnz, _hip.ptr(self._colptr), _hip.ptr(self._rowidx), _hip.ptr(self._vals_csc),
                _hip.ptr(self._rowptr), _hip.ptr(self._colidx), _hip.ptr(self._vals_csr), _hip.ptr(self._c_dev), _hip.ptr(self._lo),
                _hip.ptr(self._hi), _hip.ptr(inv_norm), _hip.dtype_code(self.dtype),
            )
        _hip.check(rc)
        return handle

    def __del__(self):
        lib = getattr(self, "_lib", None)
        for name in ("_handle_raw", "_handle"):
            h = getattr(self, name, None)
            if lib is not None and h is not None and h.value:
                if name == "_handle_raw" and h is getattr(self, "_handle", None):
                    continue
                try:
                    lib.dl_lp_destroy(h)
                except Exception:
                    pass

    def _construct_variable_lower_upper_bound(self):
        """Per-variable bounds, NaN where absent (reference miplib.py:111-121, which only reads ``l``/``u``; here also
        ``lower``/``upper``, and a box entry that names no bound has its operator's defaults [0, 1], box.py:7-13)."""
        lower = torch.full_like(self.c, float("nan"))
        upper = torch.full_like(self.c, float("nan"))
        for entry in self.projection_map.values():
            idx = _as_index_tensor(entry.indices, self.device)
            p = entry.proj_params
            lo, hi = _param(p, "l", "lower"), _param(p, "u", "upper")
            if entry.proj_type == "box":  # (missing keys: the operator's defaults, see _box_bounds)
                lo, hi = (None if v in (float("-inf"), float("inf")) else v for v in _box_bounds(p))
            if lo is not None:
                lower[idx] = lo
            if hi is not None:
                upper[idx] = hi
        return lower, upper

    def _bounds_for_kernel(self):
        """(lo, hi, generic): clamp bounds with infinities where absent for the entries that are point-wise bounds;
        ``generic`` lists (indices, operator) for everything else (None when the fused call covers the whole map)."""
        lo = torch.full((max(self.n, 1),), float("-inf"), dtype=self.dtype, device=self.device)
        hi = torch.full((max(self.n, 1),), float("inf"), dtype=self.dtype, device=self.device)
        seen = torch.zeros(max(self.n, 1), dtype=torch.int32, device=self.device)
        generic = []
        for entry in self.projection_map.values():
            idx = _as_index_tensor(entry.indices, self.device)
            if idx.numel() == 0:
                continue
            seen[idx] += 1
            p = entry.proj_params
            if entry.proj_type == "box":
                lo[idx], hi[idx] = _box_bounds(p)  # (a missing key = BoxProjection's default 0 / 1, box.py:12-13)
            elif entry.proj_type == "cone":
                l, u = _param(p, "lower", "l"), _param(p, "upper", "u")
                if l is not None and u is not None:
                    raise ValueError("Cone projection takes a lower or an upper bound, not both")
                if l is not None:
                    lo[idx] = l
                if u is not None:
                    hi[idx] = u
            else:
                generic.append((idx, _operator_for(entry)))
        if generic or bool((seen > 1).any()):
            # operators on index sets, or overlapping entries: apply every entry in map order, as the reference does
            generic = [(_as_index_tensor(e.indices, self.device), _operator_for(e)) for e in self.projection_map.values() if len(e.indices) > 0]
            return lo, hi, generic
        return lo, hi, None

    def _primal_buffer(self) -> torch.Tensor:
        if self._primal is None:
            self._primal = torch.empty(max(self.n, 1), dtype=self.dtype, device=self.device)[: self.n]
        return self._primal

    def _check_dual(self, dual_val: torch.Tensor) -> torch.Tensor:
        dual_val = _hip.stage(dual_val, "dual_val", self.device)
        if dual_val.dtype != self.dtype or dual_val.shape != (self.m,):
            raise ValueError(f"dual_val must be a {self.dtype} vector of length {self.m}")
        return dual_val.contiguous()

    # ------------------------------------------------------------------------------------------------------
    def calculate_packed_ptr(self, lambda_ptr: int, gamma: float, x_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[ (A x) [/ row_norms] (m) | c.x | sum x^2 ] in the internal float64 buffer, dual vector given by device address."""
        stream = _hip.stream_ptr(self.device)
        with torch.cuda.device(self.device):
            if self._generic_entries is None:
                _hip.check(self._lib.dl_lp_calculate(self._handle, lambda_ptr, float(gamma), _hip.ptr(self._packed), _hip.ptr(x_out), stream))
            else:
                x = x_out if x_out is not None else self._primal_buffer()
                _hip.check(self._lib.dl_lp_primal(self._handle, lambda_ptr, float(gamma), 0, _hip.ptr(x), stream))
                for idx, op in self._generic_entries:  # miplib.py:80-92
                    x[idx] = op(x[idx]).reshape(-1)  # (the reference fails on operators that return [k, 1]: miplib.py:90)
                _hip.check(self._lib.dl_lp_gradient(self._handle, _hip.ptr(x), _hip.ptr(self._packed), stream))
        return self._packed

    def calculate_packed(self, dual_val: torch.Tensor, gamma: float, x_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Local pass only (the building block of the column-sharded objective below)."""
        self.gamma = gamma
        return self.calculate_packed_ptr(_hip.ptr(self._check_dual(dual_val)), gamma, x_out)

    def finish(self, packed: torch.Tensor, dual_val: torch.Tensor, b_vec: torch.Tensor) -> ObjectiveResult:
        """grad = packed - b, dual objective and slack statistics from a (possibly all-reduced) packed buffer."""
        lam = self._check_dual(dual_val)
        grad = torch.empty(self.m, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.dl_dual_epilogue(
                self.m, _hip.dtype_code(self.dtype), _hip.ptr(packed), _hip.ptr(b_vec), _hip.ptr(lam), float(self.gamma), _hip.ptr(grad),
                _hip.ptr(self._scal), _hip.stream_ptr(self.device),
            )
        _hip.check(rc)
        s = self._scal.to(self.dtype)
        return ObjectiveResult(dual_gradient=grad, dual_objective=s[0], reg_penalty=s[1], primal_objective=s[2])

    def calculate(self, dual_val: torch.Tensor, gamma: float, save_primal: bool = False, **kwargs) -> ObjectiveResult:
        x_out = self._primal_buffer() if save_primal else None
        packed = self.calculate_packed(dual_val, gamma, x_out)
        res = self.finish(packed, dual_val, self.b_step)
        if save_primal:
            res.primal_var = x_out
        else:
            res.primal_objective = None
        return res if dual_val.is_cuda else _hip.result_to(res, dual_val.device)

    def invert_jacobi_precondition(self, dual_val: torch.Tensor, dual_grad: torch.Tensor):
        """Duals / gradient of the ORIGINAL rows from those of the row-normalised problem (run_solver.py:136-144):
        lambda = lambda~ / ||A_i||,  (A x - b) = g~ * ||A_i||."""
        if self.row_norms is None:
            return dual_val, dual_grad
        return dual_val / self.row_norms, dual_grad * self.row_norms

    # ------------------------------------------------------------------------------------------------------
    def _clamp_x_bound_duals(self, x_bound_duals, l_mask_exists, u_mask_exists):
        """Projection of the bound duals onto the set Lambda of the PDLP paper (reference miplib.py:123-154): only a
        lower bound -> >= 0; only an upper bound -> <= 0; neither -> 0; both -> free."""
        only_l = l_mask_exists & ~u_mask_exists
        only_u = ~l_mask_exists & u_mask_exists
        none = ~l_mask_exists & ~u_mask_exists
        out = torch.where(only_l, x_bound_duals.clamp(min=0), x_bound_duals)
        out = torch.where(only_u, x_bound_duals.clamp(max=0), out)
        return torch.where(none, torch.zeros_like(out), out)

    def calculate_convergence_bound(self, dual_val: torch.Tensor, x: torch.Tensor = None, optimal_primal_obj=None, tol: float = 1e-4):
        """PDLP stopping test without regularisation (reference miplib.py:156-230; Applegate et al. 2022, eq. 6a-6b).
        Returns (gap_upperbound, gap_lower_bound, primal_feas, dual_feas, converged)."""
        lam = self._check_dual(dual_val)
        stream = _hip.stream_ptr(self.device)
        # reduced cost r = c + A^T lambda' : the primal half with gamma = 1 and no bounds returns -(A^T lambda' + c) exactly
        neg_r = torch.empty(max(self.n, 1), dtype=self.dtype, device=self.device)[: self.n]
        with torch.cuda.device(self.device):
            _hip.check(self._lib.dl_lp_primal(self._handle, _hip.ptr(lam), 1.0, 0, _hip.ptr(neg_r), stream))
        r = -neg_r
        lam_s = lam if self._inv_norm is None else self._inv_norm * lam
        if x is None:
            x = torch.where(r >= 0, self.lower, self.upper)
            if torch.isnan(x).any():
                raise ValueError("Unbounded x.")
        x = x.to(device=self.device, dtype=self.dtype).contiguous()
        lambda_neg, lambda_pos = r.clamp(max=0.0), r.clamp(min=0.0)
        u_exists, l_exists = ~torch.isnan(self.upper), ~torch.isnan(self.lower)
        lambda_u = torch.dot(lambda_neg[u_exists], self.upper[u_exists])
        lambda_l = torch.dot(lambda_pos[l_exists], self.lower[l_exists])
        d = -torch.dot(self.b_vec, lam_s) + lambda_u + lambda_l
        p = torch.dot(self.c, x)
        gap_upperbound = (p - d).abs() / (1.0 + p.abs() + d.abs())
        if optimal_primal_obj is not None:
            opt = torch.as_tensor(optimal_primal_obj, dtype=self.dtype, device=self.device)
            gap_lower_bound = (p - opt).abs() / (1.0 + p.abs() + opt.abs())
        else:
            gap_lower_bound = torch.tensor(float("nan"))
        packed = torch.empty(self.m + 2, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(self._lib.dl_lp_gradient(self._handle_raw, _hip.ptr(x), _hip.ptr(packed), stream))
        ax_minus_b = packed[: self.m].to(self.dtype) - self.b_vec
        if self.equality_mask is None:
            violation = torch.relu(ax_minus_b)
        else:
            violation = torch.where(self.equality_mask.to(self.device), ax_minus_b.abs(), torch.relu(ax_minus_b))
        primal_feas = torch.linalg.vector_norm(violation) / (1.0 + torch.linalg.vector_norm(self.b_vec))
        x_bound_duals = self._clamp_x_bound_duals(-r, l_exists, u_exists)
        dual_feas = torch.linalg.vector_norm(r + x_bound_duals) / (1.0 + torch.linalg.vector_norm(self.c))
        converged = bool((gap_upperbound <= tol) and (primal_feas <= tol) and (dual_feas <= tol))
        return gap_upperbound, gap_lower_bound, primal_feas, dual_feas, converged


class MIPLIB2017ObjectiveFunctionDistributed(BaseObjective):
    """The generic-LP objective sharded by VARIABLES (columns of A), one process per GPU -- not in the reference, whose
    MIPLIB objective is single-device.  x_j depends on column j only and A x is a sum over columns, so the packed
    [A x | c.x | sum x^2] partials of the shards add up exactly like the matching objective's: one sum-all-reduce per
    iteration, then the identical device-side step on every rank (the maximizer's sharded route).  ``local_input_args``
    holds this rank's columns of A, entries of c and projection map (re-based to local indices) and the FULL b_vec."""

    _dualip_native = True

    def __init__(self, local_input_args: MIPLIBInputArgs, gamma: float, process_group=None, comm_backend=None):
        import torch.distributed as dist

        self._dist = dist
        self.local_objective = MIPLIB2017ObjectiveFunction(local_input_args, use_jacobi_precondition=False)
        self.gamma = gamma
        self.process_group = process_group
        self.comm_backend = comm_backend
        self._comm = None
        self.comm_fallback = None  # why there is no native exchange (communicator() returned None)
        self.equality_mask = local_input_args.equality_mask
        self.device, self.dtype, self.m = self.local_objective.device, self.local_objective.dtype, self.local_objective.m
        self.b_vec = self.local_objective.b_vec.contiguous()

    def communicator(self):
        """The C library's communicator for this objective's device (dualip_amd/utils/comm.py: one-shot P2P exchange or RCCL),
        created on first use -- a collective call.  None when neither back-end could be set up (``comm_fallback`` says why):
        the exchange then goes through torch.distributed."""
        if self._comm is None:
            from dualip_amd.utils.comm import make_communicator

            comm, why = make_communicator(self.m + 2, self.device, group=self.process_group, backend=self.comm_backend)
            self._comm, self.comm_fallback = (comm, None) if comm is not None else (False, why)
        return self._comm or None

    def _exchange(self, packed: torch.Tensor) -> torch.Tensor:
        if self._dist.is_available() and self._dist.is_initialized():
            if packed.is_cuda and self.communicator() is not None:
                return self.communicator().all_reduce_(packed)  # the ONE collective of an iteration, as for the matching objective
            self._dist.all_reduce(packed, op=self._dist.ReduceOp.SUM, group=self.process_group)
        return packed

    def calculate_packed_ptr(self, lambda_ptr: int, gamma: float = None) -> torch.Tensor:
        if gamma is not None:
            self.gamma = gamma
        return self._exchange(self.local_objective.calculate_packed_ptr(lambda_ptr, self.gamma))

    def calculate(self, dual_val: torch.Tensor, gamma: float = None, save_primal: bool = False, **kwargs) -> ObjectiveResult:
        if save_primal:
            raise NotImplementedError("save_primal=True is not supported in distributed mode (each rank holds its own variables)")
        if gamma is not None:
            self.gamma = gamma
        packed = self._exchange(self.local_objective.calculate_packed(dual_val, self.gamma))
        return self.local_objective.finish(packed, dual_val, self.b_vec)


def _operator_for(entry: ProjectionEntry):
    """The registered operator of an entry.  For box/cone, ``l``/``u`` are accepted as spellings of ``lower``/``upper``
    and NaN bounds are dropped; a box whose bound is ABSENT (NaN, or an ``l`` / ``u`` entry without the key -- _box_bounds) becomes
    the corresponding cone (clamp with a missing bound); a box that simply does not name a bound keeps the operator's default."""
    p = dict(entry.proj_params)
    kind = entry.proj_type
    if kind in ("box", "cone"):
        if kind == "box":
            lo, hi = (None if v in (float("-inf"), float("inf")) else v for v in _box_bounds(p))
        else:
            lo, hi = _param(p, "lower", "l"), _param(p, "upper", "u")
        p = {k: v for k, v in p.items() if k not in ("l", "u", "lower", "upper")}
        if lo is not None:
            p["lower"] = lo
        if hi is not None:
            p["upper"] = hi
        if kind == "box" and (lo is None or hi is None):
            kind = "cone"  # (one bound: clamp on that side; none: the identity, cone.py:6-28)
    return project(kind, **p)
