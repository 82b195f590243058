"""Matching-LP dual objective on the HIP path.

Reference: src/dualip/objectives/matching.py (MatchingInputArgs :12-22, MatchingSolverDualObjectiveFunction :37-188,
MatchingSolverDualObjectiveFunctionDistributed :191-307).  Same class names, constructor arguments, ``calculate``
signature and ObjectiveResult contents; the computation itself is one fused HIP pass over the CSC arrays
(``dl_matching_calculate``) plus an m-sized epilogue (``dl_dual_epilogue``) -- see include/dualip_hip.h.

Differences from the reference, all deliberate:
  * maps with several ProjectionEntry keys project every column with its own entry (the reference overwrites the
    other keys' columns with uninitialised memory, sparse_utils.py:177,220); columns in no entry are left unprojected;
  * ``batching`` is accepted and ignored: the kernel's wave tiles replace the power-of-two nnz buckets;
  * the distributed wrapper issues ONE sum-all-reduce of [A x | c.x | sum x^2] per call and every rank finishes the
    objective identically (the reference does 3 reduces + barrier and only rank 0 holds the result).
"""
import ctypes
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

from dualip_amd import _hip
from dualip_amd.objectives.base import BaseInputArgs, BaseObjective, ObjectiveResult
from dualip_amd.projections.base import ProjectionEntry, project  # noqa: F401  (ProjectionEntry is re-exported, as in the reference module)


@dataclass
class MatchingInputArgs(BaseInputArgs):
    """A, c: ``torch.sparse_csc`` (m x n) with identical pattern; one primal variable per stored non-zero."""

    A: torch.Tensor
    c: torch.Tensor
    projection_map: dict
    b_vec: Optional[torch.Tensor]
    equality_mask: Optional[torch.Tensor] = None


def _column_projection_table(projection_map, n: int, device):
    """Flatten {key: ProjectionEntry} into (descriptors, per-column entry id or None when one entry covers all)."""
    descs = []
    entries = list(projection_map.items())
    for _, entry in entries:
        d = project(entry.proj_type, **entry.proj_params).descriptor()  # raises ValueError like the reference
        # an operator without a kernel form: the fused pass zeroes its columns (clamp to [0, 0]); _CustomBlocks adds them back
        descs.append(d if d is not None else _hip.ProjDesc(_hip.PROJ_BOX, 0, 0.0, 0.0))
    if len(entries) == 1:
        idx = entries[0][1].indices
        if isinstance(idx, range) and idx == range(n):
            return descs, None
        if not isinstance(idx, (range, torch.Tensor)) and len(idx) == n and n > 0 and idx[0] == 0 and idx[-1] == n - 1:
            t = torch.as_tensor(idx)
            if torch.equal(t, torch.arange(n)):
                return descs, None
    col_proj = torch.full((n,), -1, dtype=torch.int32, device=device)
    for q, (_, entry) in enumerate(entries):
        idx = entry.indices
        if isinstance(idx, range):
            if len(idx):
                col_proj[idx.start : idx.stop : idx.step] = q
        else:
            t = torch.as_tensor(idx, dtype=torch.int64, device=device)
            if t.numel():
                if int(t.min()) < 0 or int(t.max()) >= n:
                    raise ValueError("projection_map index outside [0, n)")
                col_proj[t] = q
    return descs, col_proj


class _CustomBlocks:
    """Columns whose projection is a user-registered operator the kernel has no form for (such operators
    must keep working).  They go through zero-padded dense blocks, one per nnz-bucket -- the reference's
    apply_F_to_columns (sparse_utils.py:133-220) with its buckets (matching.py:87-114) -- built with torch ops on the
    device; index tensors are prepared once, a call does no host synchronisation."""

    def __init__(self, obj, colptr, rowidx, entries):
        dev = obj.device
        lengths = (colptr[1:] - colptr[:-1]).to(torch.int64)
        thresholds = [0]
        i = 1
        while 2**i <= obj.m:
            thresholds.append(2**i)
            i += 1
        thresholds.append(obj.m + 1)
        th = torch.tensor(thresholds, dtype=torch.int64, device=dev)
        self.blocks = []
        for entry in entries:
            op = project(entry.proj_type, **entry.proj_params)
            idx = entry.indices
            idx = torch.arange(idx.start, idx.stop, idx.step, device=dev) if isinstance(idx, range) else torch.as_tensor(idx, dtype=torch.int64, device=dev)
            lens = lengths[idx]
            idx, lens = idx[lens > 0], lens[lens > 0]
            if idx.numel() == 0:
                continue
            bucket = torch.bucketize(lens, th) if obj.batching else torch.zeros_like(lens)
            for bk in torch.unique(bucket).tolist():
                sel = bucket == bk
                cols, ln = idx[sel], lens[sel]
                K, L, E = int(cols.numel()), int(ln.max()), int(ln.sum())
                colpos = torch.repeat_interleave(torch.arange(K, device=dev), ln)
                offs = torch.arange(E, device=dev) - torch.repeat_interleave(torch.cumsum(ln, 0) - ln, ln)
                k = colptr[cols].to(torch.int64)[colpos] + offs
                self.blocks.append((op, k, rowidx[k].to(torch.int64), offs, colpos, L, K))

    def add(self, obj, lam, gamma, packed, x_out):
        a, c, m = obj._a_vals, obj._c_vals, obj.m
        scaled = -1.0 / gamma * lam  # matching.py:136
        for op, k, rows, offs, colpos, L, K in self.blocks:
            ak, ck = a[k], c[k]
            v = ak * scaled[rows] + (-1.0 / gamma * ck)  # :139-142
            block = torch.zeros((L, K), dtype=obj.dtype, device=obj.device)
            block[offs, colpos] = v
            xk = op(block)[offs, colpos]
            packed[:m].index_add_(0, rows, (ak * xk).to(torch.float64))
            packed[m] += (ck * xk).to(torch.float64).sum()
            packed[m + 1] += (xk.to(torch.float64) ** 2).sum()
            if x_out is not None:
                x_out[k] = xk


class MatchingSolverDualObjectiveFunction(BaseObjective):
    """Dual gradient / objective / regularisation penalty of the matching LP on one GPU.

    With ``b_vec=None`` it computes only the local partial sums (A x, c.x, gamma/2 ||x||^2) -- the building block of
    the distributed objective, as in the reference (matching.py:57-58, 179-184).

    ``batching`` is accepted for signature compatibility; the fused pass has no buckets.  It only matters together with
    ``simplex_eq_padding="reference"``: by default a ``simplex_eq`` entry is the exact projection onto
    {x >= 0, sum x = z} over each column's own entries; the reference projects inside zero-padded blocks, one per
    nnz-bucket (``batching=True``: buckets (0,2], (2,4], (4,8], ...) or one per entry (``batching=False``), so a clamped
    column that sums to less than z has its deficit spread over the block height instead of its own length
    (SURVEY.md 8a P4).  ``"reference"`` reproduces that, block heights computed from the column lengths as the
    reference does (matching.py:87-114, sparse_utils.py:185-186).
    """

    _dualip_native = True

    def __init__(self, matching_input_args: MatchingInputArgs, gamma: float, batching: bool = True, simplex_eq_padding: str = "exact",
                 use_jacobi_precondition: bool = False, row_norms: Optional[torch.Tensor] = None, column_slices: bool = True):
        A, c = matching_input_args.A, matching_input_args.c
        if A.layout != torch.sparse_csc or c.layout != torch.sparse_csc:
            raise ValueError("Both A and c must be CSC-format sparse tensors")
        if A.shape != c.shape or A.values().shape != c.values().shape:
            raise ValueError("A and c must share the same sparsity pattern")
        # CPU-resident inputs (the reference's default host_device, run_solver.py / its tests): copied to the current ROCm device; the
        # arithmetic is libdualip_hip.so's either way and calculate() hands its results back on the device of the duals it was given.
        # Both tensors on the host (what the reference's drivers hand over: run_matching_benchmark_dist.py:95-110): the ARRAYS the kernel
        # needs are staged through dl_stage_to_device -- pinned, chunked, several DMA queues; the int64 row indices narrowed on the host to the
        # 16 / 32 bits the handle stores anyway, so they cross the link at a quarter / half of their size and never sit in HBM as int64; an index
        # array that A and c share crosses once -- and ``self.A`` / ``self.c`` stay the CALLER's tensors (as in the reference, whose objective
        # keeps what it was given).  Maps with user-defined operators and Jacobi preconditioning work on whole device tensors: torch's copy.
        self._host_arrays = None
        if not A.values().is_cuda or not c.values().is_cuda:
            dev = _hip.compute_device() if not (A.values().is_cuda or c.values().is_cuda) else (A.values().device if A.values().is_cuda else c.values().device)
            needs_tensors = use_jacobi_precondition or os.environ.get("DUALIP_HOST_STAGING", "native") == "torch" or any(
                project(e.proj_type, **e.proj_params).descriptor() is None for e in matching_input_args.projection_map.values())
            if not (A.values().is_cuda or c.values().is_cuda) and not needs_tensors and A.values().numel() > 0:
                _hip.note_staging("A", A.values().device, dev)
                m_rows = int(A.shape[0])
                rows_src = A.row_indices()
                narrow = torch.uint16 if m_rows <= 65536 else (torch.int32 if (rows_src.dtype == torch.int64 and m_rows < 2**31) else None)
                try:
                    self._host_arrays = {
                        "colptr": _hip.stage_array(A.ccol_indices(), dev, what="ccol_indices"),
                        "rows": _hip.stage_array(rows_src, dev, narrow_to=narrow, what="row_indices"),
                        "row_code": _hip.DL_U16 if narrow == torch.uint16 else (_hip.DL_I32 if (narrow == torch.int32 or rows_src.dtype == torch.int32) else _hip.DL_I64),
                        "a": _hip.stage_array(A.values(), dev, what="A.values"),
                        "c": _hip.stage_array(c.values(), dev, what="c.values"),
                    }
                except (RuntimeError, MemoryError) as exc:  # (a HIP error, e.g. no pinned memory to be had: the pageable copy still works -- said, not silent;
                    # a ValueError -- an index that does not fit -- is the caller's and is raised)
                    import warnings

                    warnings.warn(f"dl_stage_to_device failed ({exc}); staging the host tensors with torch's pageable copy instead")
                    self._host_arrays = None
            if self._host_arrays is None:
                A, c = _hip.stage(A, "A", dev), _hip.stage(c, "c", dev)
        compute_dev = self._host_arrays["a"].device if self._host_arrays is not None else A.values().device
        b_in = matching_input_args.b_vec
        if b_in is not None and b_in.device != compute_dev:
            b_in = _hip.stage(b_in, "b_vec", compute_dev) if not b_in.is_cuda else b_in.to(compute_dev)
        # Jacobi pre-conditioning (run_solver.py:136-144 expects the objective to carry ``use_jacobi_precondition`` and
        # ``invert_jacobi_precondition``; preprocessing/precondition.py:8-28): rows of A and b scaled by 1 / ||A_i||_2 -- on
        # COPIES, the caller's tensors stay as they are.  ``row_norms`` given = the norms of the WHOLE matrix when this
        # objective holds only a column shard of it (the distributed objective all-reduces the squares).
        self.use_jacobi_precondition = bool(use_jacobi_precondition)
        self.row_norms = None
        if self.use_jacobi_precondition:
            from dualip_amd.preprocessing.precondition import jacobi_precondition
            from dualip_amd.utils.sparse_utils import left_multiply_sparse

            A = torch.sparse_csc_tensor(A.ccol_indices(), A.row_indices(), A.values().clone(), size=A.shape)
            if row_norms is None:
                b_scaled = b_in.clone() if b_in is not None else torch.ones(A.shape[0], dtype=A.values().dtype, device=A.values().device)
                self.row_norms = jacobi_precondition(A, b_scaled)
                b_in = b_scaled if b_in is not None else None
            else:
                self.row_norms = row_norms.to(device=A.values().device, dtype=A.values().dtype)
                left_multiply_sparse(1 / self.row_norms, A, A)
                b_in = b_in / self.row_norms if b_in is not None else None
        self.A, self.c = A, c
        self.gamma = gamma
        self.b_vec = b_in
        self.projection_map = matching_input_args.projection_map
        self.is_distributed = self.b_vec is None
        self.equality_mask = matching_input_args.equality_mask
        self.batching = batching
        self.device = compute_dev
        self.dtype = A.values().dtype
        self.m, self.n = int(A.shape[0]), int(A.shape[1])
        self.nnz = int(A.values().shape[0])
        if self.b_vec is not None:
            if self.b_vec.dtype != self.dtype:
                raise ValueError("b_vec must have the dtype of A")

        # keep the tensors the kernel reads alive and contiguous (values are referenced, not copied)
        if self._host_arrays is not None:  # (the staged copies; self.A / self.c are the caller's host tensors)
            self._a_vals, self._c_vals = self._host_arrays["a"], self._host_arrays["c"]
            colptr, rowidx, row_code = self._host_arrays["colptr"], self._host_arrays["rows"], self._host_arrays["row_code"]
        else:
            self._a_vals = A.values()
            self._c_vals = c.values()
            colptr = A.ccol_indices().contiguous()
            rowidx = A.row_indices().contiguous()
            row_code = _hip.idx_code(rowidx.dtype)
        if not (self._a_vals.is_contiguous() and self._c_vals.is_contiguous()):
            raise ValueError("CSC value arrays must be contiguous")
        if self._c_vals.dtype != self.dtype:
            raise ValueError("A and c must have the same dtype")
        descs, col_proj = _column_projection_table(self.projection_map, self.n, self.device)
        if not column_slices:  # keep every entry in window tiles (include/dualip_hip.h: DL_PROJ_FLAG_NO_SLICES)
            descs = [_hip.ProjDesc(d.kind, d.flags | _hip.PROJ_FLAG_NO_SLICES, d.p0, d.p1) for d in descs]
        self._descs = (_hip.ProjDesc * max(len(descs), 1))(*descs)

        lib = _hip.load()
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = lib.dl_matching_create2(
                ctypes.byref(handle),
                self.m,
                self.n,
                self.nnz,
                _hip.ptr(colptr),
                _hip.idx_code(colptr.dtype),
                _hip.ptr(rowidx),
                row_code,
                _hip.ptr(self._a_vals),
                _hip.ptr(self._c_vals),
                _hip.dtype_code(self.dtype),
                self._descs,
                len(descs),
                _hip.ptr(col_proj),
                _hip.stream_ptr(self.device),
            )
        _hip.check(rc)
        self._handle = handle
        self._lib = lib
        if self._host_arrays is not None:  # the index arrays were consumed by the handle (it owns re-encoded rows and its tile tables)
            self._host_arrays = {"a": self._a_vals, "c": self._c_vals}
        self._packed = torch.zeros(self.m + 2, dtype=torch.float64, device=self.device)
        self._scal = torch.zeros(6, dtype=torch.float64, device=self.device)
        self._primal = None  # allocated on the first save_primal, then reused (the reference aliases its scratch too)
        custom = [e for e in self.projection_map.values() if project(e.proj_type, **e.proj_params).descriptor() is None]
        self._custom = _CustomBlocks(self, colptr, rowidx, custom) if custom else None
        if self._custom is not None:  # the optimiser must hand over the duals as a tensor, one call per iteration
            self._dualip_packed = True
            self._needs_dual_tensor = True
        if simplex_eq_padding not in ("exact", "reference"):
            raise ValueError("simplex_eq_padding must be 'exact' or 'reference'")
        self.simplex_eq_padding = simplex_eq_padding
        if simplex_eq_padding == "reference" and any(e.proj_type == "simplex_eq" for e in self.projection_map.values()):
            heights = self._padded_block_heights(colptr).cpu().contiguous()
            with torch.cuda.device(self.device):
                _hip.check(lib.dl_matching_set_eq_padding(handle, heights.data_ptr(), heights.shape[0], _hip.stream_ptr(self.device)))

    def _padded_block_heights(self, colptr: torch.Tensor) -> torch.Tensor:
        """int32 [n_entries, 32]: height of the reference's zero-padded block for every (entry, nnz-bucket)."""
        lengths = (colptr[1:] - colptr[:-1]).to(torch.int64)
        thresholds = [0]
        i = 1
        while 2**i <= self.m:                      # matching.py:93-99
            thresholds.append(2**i)
            i += 1
        thresholds.append(self.m + 1)
        th = torch.tensor(thresholds, dtype=torch.int64, device=self.device)
        out = torch.zeros((len(self.projection_map), 32), dtype=torch.int64, device=self.device)
        for q, entry in enumerate(self.projection_map.values()):
            if entry.proj_type != "simplex_eq":
                continue
            idx = entry.indices
            idx = torch.arange(idx.start, idx.stop, idx.step, device=self.device) if isinstance(idx, range) else torch.as_tensor(idx, dtype=torch.int64, device=self.device)
            lens = lengths[idx]
            lens = lens[lens > 0]
            if lens.numel() == 0:
                continue
            if self.batching:
                bucket = torch.bucketize(lens, th).clamp_(max=31)        # matching.py:104, right=False
                out[q].scatter_reduce_(0, bucket, lens, reduce="amax")   # L = longest column of the bucket (sparse_utils.py:186)
            else:
                out[q, :] = lens.max()                                   # one block per entry
        return out.to(torch.int32)

    # ------------------------------------------------------------------------------------------------------
    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                self._lib.dl_matching_destroy(h)
            except Exception:
                pass
            self._handle = None

    def info(self) -> dict:
        """Kernel-side layout facts (tiles, workgroups, LDS plan) for benchmarks and tests."""
        names = ["tiles", "workgroups", "lds_bytes", "lambda_in_lds", "grad_in_lds", "owned_bytes", "long_columns", "row_index_bytes", "layout", "hot_rows", "hot_nnz_ppm", "workgroup_columns", "slices", "slice_columns", "slice_elements", "slice_nnz", "window_descriptor_words", "slice_mixed_columns"]
        info = {k: int(self._lib.dl_matching_info(self._handle, i)) for i, k in enumerate(names)}
        info["lambda_rows_in_lds"] = int(self._lib.dl_matching_info(self._handle, 2003))  # hot-rows plan: >= hot_rows; = m when the whole dual vector is staged
        info["second_binary"] = int(self._lib.dl_matching_info(self._handle, 2004))  # launches take the fused kernel's second binary (K-lane / in-place slices, dynamic deal)
        info["slice_balance_ppm"] = int(self._lib.dl_matching_info(self._handle, 2005))  # share of the one-lane slices dealt only to the early-finishing half of the workgroups (-1: even deal)
        info["slice_balance_updates"] = int(self._lib.dl_matching_info(self._handle, 2006))
        info["slab_bytes"] = int(self._lib.dl_matching_info(self._handle, 2007))  # per element of the per-workgroup gradient slabs (4: 32-bit fixed point)
        info["cold_per_xcd"] = int(self._lib.dl_matching_info(self._handle, 2009))  # hot-rows plan: per-XCD cold-row accumulators (self-checked) in use
        info["slab_rows_ok"] = int(self._lib.dl_matching_info(self._handle, 2010))  # 1: the one grid of 32-bit slabs is fine enough for every row (0 + slab_bytes 8: refused for that)
        info["slab_wide_rows"] = int(self._lib.dl_matching_info(self._handle, 2011))  # 32-bit slabs: rows (the few largest) whose high words every workgroup sends in every launch
        info["slab_overflows"] = int(self._lib.dl_matching_info(self._handle, 2008))  # workgroups that sent high words too in the last launch
        mask = int(self._lib.dl_matching_info(self._handle, 2100))  # plan switches honoured at creation (DUALIP_HIP_*)
        info["switches"] = [self._lib.dl_switch_name(i).decode() for i in range(32) if (mask >> i) & 1 and self._lib.dl_switch_name(i)]
        info["developer_build"] = int(self._lib.dl_matching_info(self._handle, 2101))
        info["slice_lane_columns"] = int(self._lib.dl_matching_info(self._handle, 2000))  # columns dealt to K = 2 .. 32 lanes each (25 .. 512 non-zeros; a handle's few short columns join them)
        return info

    def invert_jacobi_precondition(self, dual_val: torch.Tensor, dual_grad: torch.Tensor):
        """Duals / gradient of the ORIGINAL rows from those of the row-normalised problem (run_solver.py:136-144):
        lambda = lambda~ / ||A_i||,  (A x - b) = g~ * ||A_i||."""
        if self.row_norms is None:
            return dual_val, dual_grad
        return dual_val / self.row_norms, dual_grad * self.row_norms

    def release_inputs(self) -> dict:
        """Make the kernel handle self-contained and drop this objective's references to ``A`` and ``c`` (include/dualip_hip.h:
        dl_matching_own_inputs).  The reference's objective keeps both tensors for its lifetime (matching.py:79-85) -- 24 bytes per
        non-zero with torch's int64 indices -- although, here, columns held in column-per-lane slices are never read from them
        again.  After this call the handle owns the prefix of the value / row arrays its tiles read in place and NOTHING of the
        caller's: once the caller drops its own references too, the resident footprint is the handle's ``owned_bytes`` (about the
        bytes one launch streams).  Results are bit-identical.  ``values_changed`` / ``costs_changed`` are refused afterwards, and
        ``objective.A`` / ``.c`` become None.  Returns {"owned_bytes", "kept_elements"}."""
        if self._custom is not None:
            raise NotImplementedError("objectives with user-defined projection operators read A and c with torch ops every iteration")
        with torch.cuda.device(self.device):
            _hip.check(self._lib.dl_matching_own_inputs(self._handle, _hip.stream_ptr(self.device)))
        self.A = self.c = self._a_vals = self._c_vals = None
        return {"owned_bytes": int(self._lib.dl_matching_info(self._handle, 5)), "kept_elements": int(self._lib.dl_matching_info(self._handle, 2002))}

    def values_changed(self) -> None:
        """Tell the kernel handle that the values of ``A`` (and possibly ``c``) were rewritten in place, pattern unchanged.

        CONTRACT (unlike the reference, which re-reads its tensors in every call): the handle BORROWS ``A.values()`` /
        ``c.values()`` -- window tiles read them in every launch -- and OWNS transposed copies of the values of the columns it
        keeps in column-per-lane slices, plus max |a| / max |c| for its fixed-point scales.  After an in-place change call
        ``values_changed()`` (A, or both) or ``costs_changed()`` (c only) before the next ``calculate``; a changed sparsity
        pattern needs a new objective."""
        if self._host_arrays is not None:  # (host-resident caller: the device copies follow its arrays first)
            self._a_vals.copy_(self.A.values())
            self._c_vals.copy_(self.c.values())
        with torch.cuda.device(self.device):
            _hip.check(self._lib.dl_matching_update_values(self._handle, _hip.stream_ptr(self.device)))

    def costs_changed(self) -> None:
        """Tell the kernel handle that the values of ``c`` were rewritten in place (same pattern): it refreshes what it
        derived from them.  ``A`` must stay as it was when the objective was built."""
        if self._host_arrays is not None:
            self._c_vals.copy_(self.c.values())
        with torch.cuda.device(self.device):
            _hip.check(self._lib.dl_matching_update_costs(self._handle, _hip.stream_ptr(self.device)))

    def profile(self, enable) -> None:
        """Bracket fused-pass launches with HIP events on the launch stream (measurement hook): True = every launch, an
        integer N > 1 = every N-th launch, False = off."""
        _hip.check(self._lib.dl_matching_profile(self._handle, int(enable)))

    def profile_read(self):
        """(launches, total milliseconds) of the fused pass since ``profile(True)``; waits for the last launch."""
        ms, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
        _hip.check(self._lib.dl_matching_profile_read(self._handle, ctypes.byref(ms), ctypes.byref(cnt)))
        return int(cnt.value), float(ms.value)

    def timeline(self):
        """Developer aid (DUALIP_HIP_TIMELINE=1 at construction): uint64[n_wg, 8] 100 MHz stamps of the last fused launch -- columns 0 start,
        1 prologue done, 2 loop done, 3 end, 4 optimiser step derived (launches that carry it), 5 dual rows staged; 0 where a stamp was not taken."""
        import numpy as np

        n_wg = self.info()["workgroups"]
        out = np.zeros((n_wg, 8), dtype=np.uint64)
        _hip.check(self._lib.dl_matching_timeline_read(self._handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), out.size))
        return out

    def _primal_buffer(self) -> torch.Tensor:
        if self._primal is None:
            self._primal = torch.empty(self.nnz, dtype=self.dtype, device=self.device)
        return self._primal

    def _check_dual(self, dual_val: torch.Tensor) -> torch.Tensor:
        dual_val = _hip.stage(dual_val, "dual_val", self.device)
        if dual_val.dtype != self.dtype or dual_val.shape != (self.m,):
            raise ValueError(f"dual_val must be a {self.dtype} vector of length {self.m}")
        return dual_val.contiguous()

    def calculate_packed(self, dual_val: torch.Tensor, gamma: float = None, x_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Local pass only: returns the internal float64 buffer [A x (m) | c.x | sum x^2] (overwritten by the next call)."""
        lam = self._check_dual(dual_val)
        packed = self._fused_pass(_hip.ptr(lam), gamma, x_out)
        if self._custom is not None:
            self._custom.add(self, lam, self.gamma, packed, x_out)
        return packed

    def calculate_packed_ptr(self, lambda_ptr: int, gamma: float = None, x_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Same, with the dual vector given as a raw device address (the optimiser state lives inside the C library)."""
        if self._custom is not None:
            raise RuntimeError("a map with user-defined operators needs the dual vector as a tensor: use calculate_packed")
        return self._fused_pass(lambda_ptr, gamma, x_out)

    def _fused_pass(self, lambda_ptr: int, gamma: float = None, x_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if gamma is not None and gamma != self.gamma:
            self.gamma = gamma
        with torch.cuda.device(self.device):
            rc = self._lib.dl_matching_calculate(
                self._handle, lambda_ptr, float(self.gamma), _hip.ptr(self._packed), _hip.ptr(x_out), _hip.stream_ptr(self.device)
            )
        _hip.check(rc)
        return self._packed

    def finish(self, packed: torch.Tensor, dual_val: torch.Tensor, b_vec: torch.Tensor) -> ObjectiveResult:
        """grad = A x - b, dual objective and slack statistics from a (possibly all-reduced) packed buffer."""
        lam = self._check_dual(dual_val)
        grad = torch.empty(self.m, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.dl_dual_epilogue(
                self.m,
                _hip.dtype_code(self.dtype),
                _hip.ptr(packed),
                _hip.ptr(b_vec),
                _hip.ptr(lam),
                float(self.gamma),
                _hip.ptr(grad),
                _hip.ptr(self._scal),
                _hip.stream_ptr(self.device),
            )
        _hip.check(rc)
        s = self._scal.to(self.dtype)
        return ObjectiveResult(
            dual_gradient=grad,
            dual_objective=s[0],
            reg_penalty=s[1],
            dual_val_times_grad=s[3],
            max_pos_slack=s[4],
            sum_pos_slack=s[5],
        )

    def calculate(self, dual_val: torch.Tensor, gamma: float = None, save_primal: bool = False, **kwargs) -> ObjectiveResult:
        x_out = self._primal_buffer() if save_primal else None
        packed = self.calculate_packed(dual_val, gamma, x_out)
        if not self.is_distributed:
            res = self.finish(packed, dual_val, self.b_vec)
        else:
            res = ObjectiveResult(
                dual_gradient=packed[: self.m].to(self.dtype),
                dual_objective=packed[self.m].to(self.dtype),
                reg_penalty=(packed[self.m + 1] * (self.gamma / 2)).to(self.dtype),
            )
        if save_primal:
            res.primal_var = x_out
            res.primal_objective = packed[self.m].to(self.dtype)
        return res if dual_val.is_cuda else _hip.result_to(res, dual_val.device)  # (a CPU caller gets CPU results)


class MatchingSolverDualObjectiveFunctionDistributed(BaseObjective):
    """Column-sharded objective: one process per GPU, each holding a contiguous block of entities.

    Same constructor as the reference (matching.py:218-225).  ``calculate`` runs the local fused pass, sum-all-reduces
    the packed [A x | c.x | sum x^2] buffer once and finishes the objective on every rank, so all ranks can apply the
    identical dual update without a broadcast.  On the GPU the exchange is the C library's (``dl_comm``: a one-shot P2P
    exchange over hipIpc-mapped mailboxes, or RCCL -- dualip_amd/utils/comm.py); ``torch.distributed`` of the given
    ``process_group`` is the side channel that sets it up, and the exchange itself only for CPU tensors (tests drive the
    exchange logic on CPU/gloo with an oracle-backed ``local_objective``).

    ``local_matching_input_args`` may be a list of MatchingInputArgs: the rank's shard split into blocks of columns, each
    with its own kernel handle, run back to back per iteration (with RCCL the collective of every block but the last
    overlaps the next block's fused pass -- see dl_agd_run_matching_sharded).
    """

    _dualip_native = True

    def __init__(
        self,
        local_matching_input_args,
        b_vec: torch.Tensor,
        gamma: float,
        host_device=None,
        batching: bool = True,
        local_objective=None,
        process_group=None,
        comm_backend: Optional[str] = None,
        use_jacobi_precondition: bool = False,
    ):
        self.gamma = gamma
        self.use_jacobi_precondition = bool(use_jacobi_precondition) and local_objective is None
        self.row_norms = None
        self.host_device = host_device
        blocks_args = list(local_matching_input_args) if isinstance(local_matching_input_args, (list, tuple)) else [local_matching_input_args]
        first = blocks_args[0]
        self.equality_mask = first.equality_mask if first is not None else None
        self.process_group = process_group
        self.comm_backend = comm_backend
        self._comm = None
        self.comm_fallback = None  # why there is no native exchange (communicator() returned None)
        self.more_blocks = []
        if local_objective is None:
            for args in blocks_args:
                if args.b_vec is not None:
                    raise ValueError("local partitions must be built with b_vec=None (b_vec is shared by all ranks)")
            kw = {}
            if self.use_jacobi_precondition:  # ||A_i||^2 summed over every rank's (and block's) columns
                from dualip_amd.utils.sparse_utils import row_norms_csc

                sq = sum(row_norms_csc(a.A).double() ** 2 for a in blocks_args)
                if dist.is_available() and dist.is_initialized():
                    dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=process_group)
                self.row_norms = sq.sqrt().to(first.A.values().dtype)
                kw = dict(use_jacobi_precondition=True, row_norms=self.row_norms)
            local_objective = MatchingSolverDualObjectiveFunction(first, gamma, batching, **kw)
            self.more_blocks = [MatchingSolverDualObjectiveFunction(args, gamma, batching, **kw) for args in blocks_args[1:]]
            if len(self.more_blocks) > 3:
                raise ValueError("a shard can be split into at most 4 blocks")
        self.local_objective = local_objective
        self._needs_dual_tensor = any(bool(getattr(o, "_needs_dual_tensor", False)) for o in [local_objective] + self.more_blocks)
        self.device = local_objective.device
        self.dtype = local_objective.dtype
        self.m = local_objective.m
        # every rank finishes the objective on its own device (the reference moves b to host_device = cuda:0)
        self.b_vec = b_vec.to(device=self.device, dtype=self.dtype)
        if self.row_norms is not None:
            self.b_vec = self.b_vec / self.row_norms

    def invert_jacobi_precondition(self, dual_val: torch.Tensor, dual_grad: torch.Tensor):
        if self.row_norms is None:
            return dual_val, dual_grad
        return dual_val / self.row_norms, dual_grad * self.row_norms

    # ---- the exchange ---------------------------------------------------------------------------------------
    def communicator(self):
        """The C library's communicator for this objective's device (dualip_amd/utils/comm.py: one-shot P2P exchange or RCCL),
        created on first use -- a collective call.  None when neither back-end could be set up (``comm_fallback`` says why):
        the exchange then goes through torch.distributed."""
        if self._comm is None:
            from dualip_amd.utils.comm import make_communicator

            comm, why = make_communicator(self.m + 2, self.device, group=self.process_group, backend=self.comm_backend)
            self._comm, self.comm_fallback = (comm, None) if comm is not None else (False, why)
        return self._comm or None

    def block_handles(self):
        """ctypes array of the kernel handles of this rank's blocks (for dl_agd_run_matching_sharded)."""
        objs = [self.local_objective] + self.more_blocks
        return (ctypes.c_void_p * len(objs))(*[o._handle for o in objs]), len(objs)

    def _exchange(self, packed: torch.Tensor) -> torch.Tensor:
        if packed.is_cuda and hasattr(self.local_objective, "_handle") and self.communicator() is not None:
            return self.communicator().all_reduce_(packed)  # the ONE collective of an iteration
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.process_group)
        return packed

    def calculate_packed(self, dual_val: torch.Tensor, gamma: float = None, x_out=None) -> torch.Tensor:
        if gamma is not None and gamma != self.gamma:
            self.gamma = gamma
        packed = self.local_objective.calculate_packed(dual_val, self.gamma, x_out)
        for blk in self.more_blocks:
            packed += blk.calculate_packed(dual_val, self.gamma)
        return self._exchange(packed)

    def calculate_packed_ptr(self, lambda_ptr: int, gamma: float = None) -> torch.Tensor:
        if gamma is not None and gamma != self.gamma:
            self.gamma = gamma
        packed = self.local_objective.calculate_packed_ptr(lambda_ptr, self.gamma)
        for blk in self.more_blocks:
            packed += blk.calculate_packed_ptr(lambda_ptr, self.gamma)
        return self._exchange(packed)

    def calculate(self, dual_val: torch.Tensor, gamma: float = None, save_primal: bool = False, rank: int = 0, **kwargs) -> ObjectiveResult:
        if save_primal:
            raise NotImplementedError("save_primal=True is not yet supported in distributed mode")
        packed = self.calculate_packed(dual_val, gamma)
        self.local_objective.gamma = self.gamma
        res = self.local_objective.finish(packed, dual_val, self.b_vec)
        return res if (dual_val.is_cuda or self.device.type != "cuda") else _hip.result_to(res, dual_val.device)
