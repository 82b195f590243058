"""Matching LP with a fairness constraint between two groups of entities.

Reference: the extension worked through in docs/demo/matching_complex.rst:8-168 -- the dual vector grows by two rows,
``b' = (b_1..b_K, delta, delta)``, whose coefficients are a scaled copy of A's values (``A_fairness``: +A/|T1| on the
first group's columns, -A/|T2| on the rest, :45-63) entering every non-zero:

    v_k    = a_k s[r_k] + f_k s[K] - f_k s[K+1] + c_k (-1/gamma),      s = -lambda / gamma          (:103-118)
    grad_K = sum_k f_k x_k ,  grad_{K+1} = -sum_k f_k x_k                                             (:128-130)

Two implementations, chosen by ``native`` (default: the first when the kernel layout allows it):

NATIVE -- ``dl_matching_set_fairness`` (include/dualip_hip.h): the fused kernel streams f beside a and c (16 instead of 12
bytes per non-zero in fp32), adds ``f_k (s[K] - s[K+1])`` to v_k and returns ``sum f_k x_k`` as rows K / K+1 of A x.  The
objective is then an ordinary matching objective with K + 2 rows: the whole AGD loop stays on the device, any projection
map works, nothing is rewritten per iteration.

FOLDED -- fallback without kernel support (staged inputs -- unaligned or tiny value arrays --, dual vector not in LDS): the two dense rows are folded into
the cost the fused pass sees.  With d = lambda_K - lambda_{K+1},

    v_k = a_k s[r_k] + (-1/gamma) (c_k + d f_k)

so one launch of the ordinary fused matching kernel on the per-iteration cost ``c + d f`` returns x, A x, sum x^2 and
(c + d f).x; one dot product gives F = f.x, and c.x = (c + d f).x - d F.  Per iteration that is one element-wise pass,
the fused pass with the primal written out, and a dot product: ~36 bytes per non-zero against 12 for the plain
objective.

In the folded form the projection map must bound x (box with both bounds, simplex, simplex_eq): the kernel's fixed-point
gradient scale is derived from max|c| only for unbounded projections, and the cost changes every iteration there.
The reference ships no runnable code for this extension; tests compare against oracle/fairness_oracle.py and against a
fixture produced by composing the reference's own sparse operators (tests/golden/make_golden_fair.py).
"""
from typing import Optional

import torch

from dualip_amd import _hip
from dualip_amd.objectives.base import BaseObjective
from dualip_amd.objectives.matching import MatchingInputArgs, MatchingSolverDualObjectiveFunction
from dualip_amd.types import ObjectiveResult

_BOUNDED = ("simplex", "simplex_eq")


def fairness_coefficients(A: torch.Tensor, group_ratio: float) -> torch.Tensor:
    """Values of ``A_fairness`` in A's non-zero order: A/|T1| on the first ``int(n * group_ratio)`` columns, -A/|T2| on
    the remaining ones (matching_complex.rst:47-61).  An empty group contributes nothing."""
    n = int(A.shape[1])
    n1 = max(0, min(int(n * group_ratio), n))
    n2 = n - n1
    colptr = A.ccol_indices()
    split = int(colptr[n1])
    vals = A.values()
    f = torch.empty_like(vals)
    if n1 > 0:
        torch.mul(vals[:split], 1.0 / n1, out=f[:split])
    if n2 > 0:
        torch.mul(vals[split:], -1.0 / n2, out=f[split:])
    return f


class MatchingFairnessDualObjectiveFunction(BaseObjective):
    """``matching_input_args.A`` / ``c`` are the K x n matching problem, ``b_vec`` has K + 2 entries (the last two are the
    tolerance delta), duals have K + 2 entries.  ``A_fairness`` may be given as a CSC tensor with A's pattern or as a
    values tensor; by default it is built from ``group_ratio`` as the reference's demo does."""

    _dualip_native = True
    _dualip_packed = True

    def __init__(self, matching_input_args: MatchingInputArgs, gamma: float, group_ratio: float = 0.5, A_fairness: Optional[torch.Tensor] = None, batching: bool = True,
                 native: Optional[bool] = None):
        if not matching_input_args.A.values().is_cuda:  # CPU-resident inputs: staged to the current ROCm device (dualip_amd/_hip.py: stage)
            dev = _hip.note_staging("A", matching_input_args.A.values().device)  # (says it once; raises without a GPU; copies nothing: .to() below does)
            matching_input_args = matching_input_args.to(dev)
            A_fairness = None if A_fairness is None else A_fairness.to(dev)
        A, c = matching_input_args.A, matching_input_args.c
        self.k = int(A.shape[0])
        self.m = self.k + 2
        b = matching_input_args.b_vec
        if b is None or b.shape != (self.m,):
            raise ValueError(f"b_vec must have {self.m} entries: the {self.k} row limits followed by the two fairness tolerances")
        if A_fairness is None:
            f = fairness_coefficients(A, group_ratio)
        else:
            f = A_fairness.values() if A_fairness.layout == torch.sparse_csc else A_fairness
            if f.shape != A.values().shape or f.dtype != A.values().dtype or f.device != A.values().device:
                raise ValueError("A_fairness must share A's sparsity pattern, dtype and device")
        self._f = f.contiguous()
        self._lib = _hip.load()
        self.native = False
        if native is None or native:
            if self._try_native(matching_input_args, gamma, batching):
                return
            if native:
                raise ValueError("the fused kernel cannot take the fairness stream for this input: " + _hip.last_error())
        for entry in matching_input_args.projection_map.values():
            p = entry.proj_params or {}
            two_sided = entry.proj_type == "box" and (not p or (("lower" in p or "l" in p) and ("upper" in p or "u" in p)))
            if entry.proj_type not in _BOUNDED and not two_sided:
                raise NotImplementedError(f"the fairness objective needs projections that bound x; got {entry.proj_type} {p}")
        self._c0 = c.values()
        self._c_eff = self._c0.clone()
        c_eff = torch.sparse_csc_tensor(c.ccol_indices(), c.row_indices(), self._c_eff, size=c.shape)
        if c_eff.values().data_ptr() != self._c_eff.data_ptr():  # the kernel must see the buffer this class rewrites
            self._c_eff = c_eff.values()
        inner_args = MatchingInputArgs(A=A, c=c_eff, projection_map=matching_input_args.projection_map, b_vec=None, equality_mask=None)
        self.inner = MatchingSolverDualObjectiveFunction(matching_input_args=inner_args, gamma=gamma, batching=batching, column_slices=False)
        self.gamma = gamma
        self.b_vec = b
        self.equality_mask = matching_input_args.equality_mask
        self.device, self.dtype = self.inner.device, self.inner.dtype
        self.nnz = self.inner.nnz
        self._packed = torch.zeros(self.m + 2, dtype=torch.float64, device=self.device)
        self._scal = torch.zeros(8, dtype=torch.float64, device=self.device)

    def _try_native(self, args: MatchingInputArgs, gamma: float, batching: bool) -> bool:
        """The same arrays as a (K + 2) x n problem whose last two rows are the pair (no copy: the CSC views share storage)."""
        A, c = args.A, args.c
        n = int(A.shape[1])
        if self._f.data_ptr() % 16:
            return False
        wide = lambda t: torch.sparse_csc_tensor(t.ccol_indices(), t.row_indices(), t.values(), size=(self.m, n))  # noqa: E731
        inner_args = MatchingInputArgs(A=wide(A), c=wide(c), projection_map=args.projection_map, b_vec=args.b_vec, equality_mask=args.equality_mask)
        # window tiles only: with a fourth streamed array the sliced kernel spills registers and is slower than the window one
        inner = MatchingSolverDualObjectiveFunction(matching_input_args=inner_args, gamma=gamma, batching=batching, column_slices=False)
        with torch.cuda.device(inner.device):
            rc = self._lib.dl_matching_set_fairness(inner._handle, _hip.ptr(self._f), _hip.stream_ptr(inner.device))
        if rc != 0:
            return False
        self.inner, self.native = inner, True
        self.gamma, self.b_vec, self.equality_mask = gamma, args.b_vec, args.equality_mask
        self.device, self.dtype, self.nnz = inner.device, inner.dtype, inner.nnz
        # the optimiser treats this object as the matching objective it wraps: device-resident loop on the same handle
        self._dualip_packed = bool(getattr(inner, "_dualip_packed", False))
        self._needs_dual_tensor = bool(getattr(inner, "_needs_dual_tensor", False))
        self._handle = inner._handle
        self.calculate_packed_ptr = inner.calculate_packed_ptr
        return True

    def _primal_buffer(self) -> torch.Tensor:
        return self.inner._primal_buffer()

    def calculate_packed(self, dual_val: torch.Tensor, gamma: float = None, x_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """float64 [A' x (K + 2) | c.x | sum x^2]; no host synchronisation."""
        if gamma is not None:
            self.gamma = gamma
        if self.native:
            return self.inner.calculate_packed(dual_val, self.gamma, x_out)
        k = self.k
        lam = dual_val.contiguous()
        d = lam[k] - lam[k + 1]
        torch.addcmul(self._c0, self._f, d, out=self._c_eff)
        self.inner.costs_changed()  # the handle keeps derived copies of the costs (include/dualip_hip.h: dl_matching_update_costs)
        x = self.inner._primal_buffer() if x_out is None else x_out
        inner = self.inner.calculate_packed_ptr(_hip.ptr(lam), self.gamma, x)
        fx = torch.dot(self._f, x).to(torch.float64)
        out = self._packed
        out[:k] = inner[:k]
        out[k] = fx
        out[k + 1] = -fx
        out[k + 2] = inner[k] - d.to(torch.float64) * fx
        out[k + 3] = inner[k + 1]
        return out

    def finish(self, packed: torch.Tensor, dual_val: torch.Tensor, b_vec: torch.Tensor) -> ObjectiveResult:
        if self.native:
            self.inner.gamma = self.gamma
            return self.inner.finish(packed, dual_val, b_vec)
        grad = torch.empty(self.m, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(
                self._lib.dl_dual_epilogue(
                    self.m, _hip.dtype_code(self.dtype), _hip.ptr(packed), _hip.ptr(b_vec), _hip.ptr(dual_val.contiguous()), float(self.gamma),
                    _hip.ptr(grad), _hip.ptr(self._scal), _hip.stream_ptr(self.device),
                )
            )
        s = self._scal.to(self.dtype)
        return ObjectiveResult(dual_gradient=grad, dual_objective=s[0], reg_penalty=s[1], dual_val_times_grad=s[3], max_pos_slack=s[4], sum_pos_slack=s[5])

    def calculate(self, dual_val: torch.Tensor, gamma: float = None, save_primal: bool = False, **kwargs) -> ObjectiveResult:
        caller = None if dual_val.is_cuda else dual_val.device
        dual_val = _hip.stage(dual_val, "dual_val", self.device)
        if dual_val.dtype != self.dtype or dual_val.shape != (self.m,):
            raise ValueError(f"dual_val must be a {self.dtype} vector of length {self.m}")
        if self.native:
            if gamma is not None:
                self.gamma = gamma
            return _hip.result_to(self.inner.calculate(dual_val, self.gamma, save_primal), caller)
        packed = self.calculate_packed(dual_val, gamma)
        res = self.finish(packed, dual_val, self.b_vec)
        if save_primal:
            res.primal_var = self.inner._primal_buffer()
            res.primal_objective = packed[self.m].to(self.dtype)
        return _hip.result_to(res, caller)
