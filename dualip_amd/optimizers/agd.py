"""Accelerated (Nesterov) projected gradient ascent on the dual.

Reference: src/dualip/optimizers/agd.py (AcceleratedGradientDescent :66-229, project_on_nn_cone :13-21,
format_objective_result_summary :24-63).  Constructor arguments, ``maximize(f, initial_value, rank=0)`` and the
returned SolverResult are those of the reference.

Three execution routes, chosen by what ``f`` is:
  1. native matching objective on one GPU  -> the whole loop runs on the device (``dl_agd_run_matching``): fused
     objective pass + step kernel per iteration, no host synchronisation; logs are fetched in chunks;
  2. native column-sharded objective       -> per iteration: local fused pass, ONE RCCL sum-all-reduce of the packed
     partial, then the same device step on every rank (identical duals on all ranks, no broadcast);
  3. any other BaseObjective (user defined) -> generic torch implementation of the same recurrences, on whatever
     device the objective's tensors live.
``iteration_callback``: None = print one summary line per iteration (reference default; routes 1/2 print from the
device log after each chunk), a callable = called every iteration with the full ObjectiveResult (forces one host
synchronisation per iteration on routes 1/2), False = silent.
"""
import ctypes
import math
import os
from typing import Callable, Optional

import numpy as np
import torch
import torch.distributed as dist

from dualip_amd import _hip
from dualip_amd.objectives.base import BaseObjective
from dualip_amd.optimizers.agd_utils import calculate_step_size
from dualip_amd.types import ObjectiveResult, SolverResult
from dualip_amd.utils.mlflow_utils import log_iteration_rows, log_metrics, log_objective_result, tracking_enabled


def project_on_nn_cone(y: torch.Tensor, equality_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Clamp the inequality duals at zero; rows flagged in ``equality_mask`` are free."""
    clamped = y.clamp(min=0)
    return clamped if equality_mask is None else torch.where(equality_mask, y, clamped)


def format_objective_result_summary(iteration: int, objective_result: ObjectiveResult) -> str:
    """``iter=… | dual_objective=… | dual_grad_norm=… | <optional fields>`` -- one line per iteration."""

    def show(name, val):
        if val is None:
            return None
        try:
            if isinstance(val, torch.Tensor):
                return f"{name}={val.item()}" if val.numel() == 1 else f"{name}.shape={tuple(val.shape)}"
            return f"{name}={val}"
        except Exception:
            return f"{name}=<unprintable>"

    try:
        norm = f"dual_grad_norm={float(objective_result.dual_gradient.norm().item())}"
    except Exception:
        norm = "dual_grad_norm=<unprintable>"
    fields = [f"iter={iteration}", show("dual_objective", objective_result.dual_objective), norm]
    for name in ("reg_penalty", "primal_objective", "primal_var", "dual_val_times_grad", "max_pos_slack", "sum_pos_slack"):
        fields.append(show(name, getattr(objective_result, name, None)))
    return " | ".join(f for f in fields if f is not None)


def compute_beta_seq(max_iter: int) -> torch.Tensor:
    """Momentum weights beta_i = (1 - t_{i+1}) / t_{i+2}, t_0 = 0, t_i = (1 + sqrt(1 + 4 t_{i-1}^2)) / 2.

    Bit-compatible with the reference (agd.py:93-100), whose float32 storage rounds every t_i: the recurrence is
    evaluated with float32 products/sums, the square root in double, and a float32 division at the end.
    """
    f32 = np.float32
    t = np.zeros(max_iter + 2, dtype=f32)
    for i in range(1, max_iter + 2):
        inner = f32(f32(1.0) + f32(f32(4.0) * f32(t[i - 1] * t[i - 1])))
        t[i] = f32((1.0 + math.sqrt(float(inner))) / 2.0)
    beta = (f32(1.0) - t[1 : max_iter + 1]) / t[2 : max_iter + 2] if max_iter > 0 else np.zeros(0, dtype=f32)
    return torch.from_numpy(np.asarray(beta, dtype=f32))


class AcceleratedGradientDescent:
    def __init__(
        self,
        max_iter: int,
        gamma: float,
        initial_step_size: float = 1e-5,
        max_step_size: float = 0.1,
        gamma_decay_type: str = None,
        gamma_decay_params: dict = {},
        save_primal: bool = False,
        iteration_callback: Optional[Callable[[int, ObjectiveResult], None]] = None,
    ):
        self.initial_step_size = initial_step_size
        self.max_step_size = max_step_size
        self.max_iter = max_iter
        self.beta_seq = compute_beta_seq(max_iter)
        self.streams = None
        self.gamma = gamma
        self.gamma_decay_type = gamma_decay_type
        self.gamma_decay_params = gamma_decay_params
        self.save_primal = save_primal
        self._user_callback = iteration_callback
        self.iteration_callback = self._default_iteration_callback if iteration_callback in (None, False) else iteration_callback
        self.log_chunk = 100  # native routes: iterations between host reads of the device log
        self.attempt = 1  # 2 while a sharded solve is being repeated after a degraded exchange (_maximize_native)

    # ---- small pieces shared by all routes --------------------------------------------------------------
    def _compute_beta_seq(self, max_iter: int) -> torch.Tensor:
        return compute_beta_seq(max_iter)

    def _check_decay(self):
        if self.gamma is not None and self.gamma_decay_type is not None and self.gamma_decay_type != "step":
            raise ValueError(f"Unsupported gamma decay type: {self.gamma_decay_type}")

    def _update_gamma(self, itr: int, step_size: float):
        if self.gamma_decay_type != "step":
            raise ValueError(f"Unsupported gamma decay type: {self.gamma_decay_type}")
        if itr % self.gamma_decay_params["decay_steps"] == 0:
            factor = self.gamma_decay_params["decay_factor"]
            self.gamma = self.gamma * factor
            self.max_step_size = step_size * factor

    def _default_iteration_callback(self, iteration: int, objective_result: ObjectiveResult) -> None:
        try:
            print(format_objective_result_summary(iteration, objective_result))
        except Exception:
            pass  # logging must never stop the solve

    # ---- entry point ----------------------------------------------------------------------------------------
    def maximize(self, f: BaseObjective, initial_value: torch.Tensor, rank: int = 0) -> SolverResult:
        if getattr(f, "_dualip_native", False) and self.gamma is not None:
            if initial_value.is_cuda:
                return self._maximize_native(f, initial_value, rank)
            if getattr(getattr(f, "device", None), "type", None) == "cuda":
                # a CPU caller of an objective that lives on the GPU (its CPU inputs were staged there, dualip_amd/_hip.py: stage): the same
                # device-resident loop, and the result's tensors on the caller's device
                return _hip.result_to(self._maximize_native(f, initial_value, rank), initial_value.device)
        return self._maximize_generic(f, initial_value, rank)

    # ---- route 3: generic torch path ---------------------------------------------------------------------------
    def _maximize_generic(self, f, initial_value, rank):
        grad_hist, dual_hist, obj_log, step_log = [], [], [], []
        x = initial_value.clone()
        y = initial_value.clone()
        mask = f.equality_mask
        result, dual_obj = None, 0.0
        for i in range(1, self.max_iter + 1):
            kw = {} if self.gamma is None else {"gamma": self.gamma}
            if i == self.max_iter and self.save_primal:
                kw["save_primal"] = True
            result = f.calculate(dual_val=x, rank=rank, **kw)
            if rank == 0:
                if self._user_callback is not False:
                    self.iteration_callback(i, result)
                dual_obj = result.dual_objective.cpu().item()
                obj_log.append(dual_obj)
                step = calculate_step_size(
                    result.dual_gradient, y, grad_hist, dual_hist, initial_step_size=self.initial_step_size, max_step_size=self.max_step_size
                )
                step_log.append(step)
                y_next = project_on_nn_cone(x + result.dual_gradient * step, mask)
                beta = self.beta_seq[i - 1]
                x = y_next * (1.0 - beta) + y * beta
                y = y_next
                if self.gamma is not None and self.gamma_decay_type is not None:
                    self._update_gamma(i, step)
                if tracking_enabled():  # (agd.py:189-201)
                    log_metrics({"step_size": float(step), "dual_objective": dual_obj, **({} if self.gamma is None else {"gamma": float(self.gamma)})}, step=i)
                    log_objective_result(result, step=i)
            if dist.is_available() and dist.is_initialized():
                dist.broadcast(x, src=0)
                dist.broadcast(y, src=0)
        if rank == 0:
            return SolverResult(dual_val=y, dual_objective=dual_obj, objective_result=result, dual_objective_log=obj_log, step_size_log=step_log)
        return SolverResult(dual_val=y, dual_objective=0.0, objective_result=result, dual_objective_log=[], step_size_log=[])

    # ---- routes 1 and 2: device-resident loop ------------------------------------------------------------------
    def start_device_run(self, f, initial_value: torch.Tensor, rank: int = 0) -> "DeviceRun":
        """Create the device-resident optimiser state for a native objective without iterating yet
        (``maximize`` = start_device_run + advance(max_iter) + finish; benchmarks time ``advance`` directly)."""
        self._check_decay()
        return DeviceRun(self, f, initial_value, rank)

    def _maximize_native(self, f, initial_value, rank):
        """The device-resident loop.  A sharded run whose exchange was chosen by ``DUALIP_COMM=auto`` gets ONE more attempt when a
        payload checksum did not match under the unfenced P2P ordering: every rank moves to the fenced ordering together
        (``Communicator.degrade``) and the solve is repeated from ``initial_value`` -- it is deterministic, so nothing of the failed
        attempt survives.  Any other failure (a timed-out wait, an explicitly chosen back-end) is raised on every rank.
        ``self.attempt`` is 1 during the first pass and 2 during the repeat: iteration callbacks, printed summaries and tracked metrics of
        the failed pass are issued AGAIN from iteration 1 by the repeat (the warning says so); a callback that must not see an iteration
        twice checks ``solver.attempt``."""
        from dualip_amd.utils.comm import CHECKSUM, ExchangeError

        state = (self.gamma, self.max_step_size)
        self.attempt = 1
        try:
            return self._maximize_native_once(f, initial_value, rank)
        except ExchangeError as exc:
            comm = f.communicator()
            if not (all(c in (0, CHECKSUM) for c in exc.codes) and comm is not None and comm.degrade()):
                raise
            import warnings

            warnings.warn(f"dualip_amd: {exc}; repeating the solve with the fenced exchange (iteration callbacks and logged metrics start again at iteration 1: solver.attempt == 2)")
            self.gamma, self.max_step_size = state
            self.attempt = 2
            return self._maximize_native_once(f, initial_value, rank)

    def _maximize_native_once(self, f, initial_value, rank):
        run = self.start_device_run(f, initial_value, rank)
        try:
            per_iteration = callable(self._user_callback)
            chunk = 1 if per_iteration else max(1, int(self.log_chunk))
            # Host-side work between chunks (callbacks, printing, a tracking server) happens on rank 0 only, while the in-kernel
            # waits of the P2P exchange are bounded: when any rank may do such work, all ranks meet on the host after every chunk
            # and check the exchange, so a slow callback delays the others instead of timing them out, and a dead exchange stops
            # the solve at once instead of burning the remaining iterations.  (The reference simply blocks in its collectives.)
            meet = run.native_sharded and _any_rank(self._user_callback is not False or tracking_enabled(), getattr(f, "process_group", None))
            while run.done < self.max_iter:
                first = run.done
                n = run.advance(min(chunk, self.max_iter - run.done))
                track = rank == 0 and tracking_enabled()
                if per_iteration or self._user_callback is None or track:
                    rows = run.read_log(first, n)
                    if track:
                        log_iteration_rows(first + 1, rows, gammas=run.gammas_after(first + 1, n), with_primal_last=self.save_primal and run.done == self.max_iter)
                    for k in range(n if (per_iteration or self._user_callback is None) else 0):
                        it = first + k + 1
                        if per_iteration:
                            if rank == 0:  # (the reference calls back on rank 0 only, agd.py:166-168)
                                self.iteration_callback(it, run.result_from_row(rows[k], with_grad=True))
                        elif rank == 0:
                            print(_summary_from_log(it, rows[k]))
                if meet:
                    f.communicator().meet()  # (a dead exchange stops the solve on EVERY rank at once)
            return run.finish()
        finally:
            run.close()


def _summary_from_log(iteration: int, row) -> str:
    return (
        f"iter={iteration} | dual_objective={row[0]} | dual_grad_norm={row[6]} | reg_penalty={row[2]} | "
        f"dual_val_times_grad={row[3]} | max_pos_slack={row[4]} | sum_pos_slack={row[5]}"
    )


def dual_digest(dual_val: torch.Tensor) -> int:
    """Order-sensitive 63-bit digest of a dual vector's BYTES (two vectors that differ in one bit differ here)."""
    bits = dual_val.detach().contiguous().view(torch.int32 if dual_val.element_size() == 4 else torch.int64).to(torch.int64)
    w = torch.arange(1, bits.numel() + 1, dtype=torch.int64, device=bits.device) * 0x9E3779B1
    return int((bits * w).sum().item()) & 0x7FFFFFFFFFFFFFFF


def _assert_same_duals(dual_val: torch.Tensor, group=None) -> None:
    """Every rank of a column-sharded solve must end with the same duals (the reference broadcasts rank 0's, agd.py:204-206;
    here every rank applies the identical update to identical sums).  A collective call at the end of every sharded ``maximize``:
    all ranks raise together when the byte digests differ -- an exchange that handed different sums to different ranks is then
    loud, whatever the back-end."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    mine = dual_digest(dual_val)
    all_d = [None] * dist.get_world_size(group)
    dist.all_gather_object(all_d, mine, group=group)
    if len(set(all_d)) != 1:
        raise RuntimeError(f"column-sharded solve: the ranks ended with DIFFERENT duals (byte digests {all_d}); the exchange handed them different sums")


def _all_ranks(flag: bool, group=None) -> bool:
    """True iff ``flag`` is true on every rank of the process group (a collective call; the answer is the same everywhere)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(flag)
    flags = [None] * dist.get_world_size(group)
    dist.all_gather_object(flags, bool(flag), group=group)
    return all(flags)


def _any_rank(flag: bool, group=None) -> bool:
    return not _all_ranks(not flag, group)


class DeviceRun:
    """One maximize() run whose x, y, gradient history, step-size ring and logs live on the GPU (``dl_agd`` in
    include/dualip_hip.h).  ``advance(n)`` enqueues n iterations and returns without synchronising."""

    def __init__(self, solver: "AcceleratedGradientDescent", f, initial_value: torch.Tensor, rank: int = 0):
        self.solver, self.f, self.rank = solver, f, rank
        self.lib = _hip.load()
        self.sharded = hasattr(f, "local_objective")
        # objectives that hand over a packed [A x | c.x | sum x^2] buffer per iteration: the sharded matching objective
        # (after its all-reduce) and the generic-LP objective; the single-device matching objective runs wholly inside C
        self.packed_route = self.sharded or bool(getattr(f, "_dualip_packed", False))
        self.local = f.local_objective if self.sharded else f
        # the column-sharded matching objective on the GPU: the whole loop, exchange included, runs inside the C library
        # (dl_agd_run_matching_sharded); objectives that need the duals as a tensor every iteration (user-defined projection
        # operators, injected local objectives) keep the per-iteration route below
        self.native_sharded = (
            self.sharded and hasattr(f, "block_handles") and hasattr(self.local, "_handle") and not getattr(f, "_needs_dual_tensor", False)
            and self.local.device.type == "cuda" and os.environ.get("DUALIP_SHARDED_LOOP", "c") != "python"
        )
        # The route is a COLLECTIVE decision: the two routes issue different collectives (the C loop's exchange against one
        # all-reduce per iteration from Python), and what decides it above is rank-local -- a map whose user-defined operator
        # covers columns of one shard only, or DUALIP_SHARDED_LOOP set on one rank, would make the ranks disagree and hang.
        # The C loop runs only if EVERY rank can take it.
        if self.sharded:
            self.native_sharded = _all_ranks(self.native_sharded, getattr(f, "process_group", None))
        if self.native_sharded and f.communicator() is None:  # (collective; no native exchange on this machine: per-iteration route)
            self.native_sharded = False
        if not self.sharded and f.b_vec is None:
            raise ValueError("a matching objective built with b_vec=None only provides local partial sums; wrap it in the distributed objective")
        self.device, self.dtype, self.m = self.local.device, self.local.dtype, self.local.m
        self.b_vec = getattr(f, "b_step", f.b_vec)  # (row-normalised b for a Jacobi-preconditioned LP objective)
        self.max_iter = solver.max_iter
        if solver.save_primal and self.sharded:
            raise NotImplementedError("save_primal=True is not yet supported in distributed mode")
        self.primal = self.local._primal_buffer() if solver.save_primal else None
        lam0 = initial_value.to(device=self.device, dtype=self.dtype).contiguous()
        mask = f.equality_mask
        self._mask_u8 = None if mask is None else mask.to(device=self.device).to(torch.uint8).contiguous()
        self.decay_steps, self.decay_factor = 0, 1.0
        if solver.gamma_decay_type == "step":
            self.decay_steps = int(solver.gamma_decay_params["decay_steps"])
            self.decay_factor = float(solver.gamma_decay_params["decay_factor"])
        self._beta_host = solver.beta_seq.contiguous()
        self.gamma = ctypes.c_double(float(solver.gamma))
        self.gamma0 = float(solver.gamma)
        self.done = 0
        self._x_dev = None
        self.state = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _hip.check(
                self.lib.dl_agd_create(
                    ctypes.byref(self.state), self.m, _hip.dtype_code(self.dtype), self.max_iter, _hip.ptr(self._beta_host),
                    float(solver.initial_step_size), float(solver.max_step_size), _hip.ptr(self._mask_u8), _hip.ptr(lam0), _hip.stream_ptr(self.device),
                )
            )

        if self.native_sharded:
            f.communicator().rendezvous()  # the in-kernel waits of the exchange are bounded: start the ranks together

    def close(self):
        if self.state is not None and self.state.value:
            self.lib.dl_agd_destroy(self.state)
        self.state = None

    def _fetch(self, which: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Copy a state vector owned by the C library (0 = x, 1 = y, 2 = gradient) into a torch tensor."""
        if out is None:
            out = torch.empty(self.m, dtype=self.dtype, device=self.device)
        _hip.check(self.lib.dl_agd_get(self.state, which, _hip.ptr(out), _hip.stream_ptr(self.device)))
        return out

    def advance(self, n: int) -> int:
        n = min(int(n), self.max_iter - self.done)
        if n <= 0:
            return 0
        lib = self.lib
        with torch.cuda.device(self.device):
            stream = _hip.stream_ptr(self.device)
            last = self.done + n == self.max_iter
            if not self.packed_route:
                _hip.check(
                    lib.dl_agd_run_matching(
                        self.state, self.local._handle, _hip.ptr(self.b_vec), self.done + 1, n, ctypes.byref(self.gamma), self.decay_steps,
                        self.decay_factor, _hip.ptr(self.primal) if (last and self.primal is not None) else None, stream,
                    )
                )
            elif self.native_sharded:
                handles, n_blocks = self.f.block_handles()
                _hip.check(
                    lib.dl_agd_run_matching_sharded(
                        self.state, handles, n_blocks, self.f.communicator().handle, _hip.ptr(self.b_vec), self.done + 1, n, ctypes.byref(self.gamma),
                        self.decay_steps, self.decay_factor, stream,
                    )
                )
            else:
                by_address = hasattr(self.f, "calculate_packed_ptr") and not getattr(self.f, "_needs_dual_tensor", False)
                for it in range(self.done + 1, self.done + n + 1):
                    extra = {"x_out": self.primal} if (self.primal is not None and it == self.max_iter and not self.sharded) else {}
                    if by_address:
                        packed = self.f.calculate_packed_ptr(int(lib.dl_agd_x(self.state)), self.gamma.value, **extra)  # local pass [+ ONE sum-all-reduce]
                    else:  # objectives that read the duals with torch ops (user-defined projection operators, fairness rows)
                        self._x_dev = self._fetch(0, out=self._x_dev)
                        packed = self.f.calculate_packed(self._x_dev, self.gamma.value, **extra)
                    decay_now = int(self.decay_steps > 0 and it % self.decay_steps == 0)
                    _hip.check(lib.dl_agd_step(self.state, _hip.ptr(packed), _hip.ptr(self.b_vec), self.gamma.value, it, decay_now, self.decay_factor, stream))
                    if decay_now:
                        self.gamma.value = self.gamma.value * self.decay_factor
        self.done += n
        return n

    def gammas_after(self, first_iteration: int, count: int):
        """gamma as the reference reports it at the end of iterations first_iteration .. +count-1 (after the decay, agd.py:186-196)."""
        if self.decay_steps <= 0:
            return [self.gamma0] * count
        return [self.gamma0 * self.decay_factor ** (it // self.decay_steps) for it in range(first_iteration, first_iteration + count)]

    def read_log(self, first: int, count: int) -> np.ndarray:
        rows = np.zeros((max(count, 0), _hip.LOG_COLS), dtype=np.float64)
        with torch.cuda.device(self.device):
            _hip.check(self.lib.dl_agd_read_log(self.state, first, count, rows.ctypes.data, _hip.stream_ptr(self.device)))
        return rows

    def result_from_row(self, row, with_grad: bool) -> ObjectiveResult:
        grad = self._fetch(2) if with_grad else torch.empty(0, dtype=self.dtype, device=self.device)
        return ObjectiveResult.from_log_row(row, grad, self.dtype, self.device)

    def finish(self) -> SolverResult:
        solver = self.solver
        if self.native_sharded:
            # a failed exchange invalidates the run: every rank raises together instead of returning numbers (Communicator.meet)
            self.f.communicator().meet()
        rows = self.read_log(0, self.done)
        mx = ctypes.c_double(0.0)
        with torch.cuda.device(self.device):
            _hip.check(self.lib.dl_agd_read_max_step(self.state, ctypes.byref(mx), _hip.stream_ptr(self.device)))
            dual_val = self._fetch(1)
            final = self.result_from_row(rows[-1], with_grad=True) if self.done > 0 else None
        if self.sharded and self.native_sharded:
            _assert_same_duals(dual_val, getattr(self.f, "process_group", None))
        solver.gamma = self.gamma.value
        if self.decay_steps > 0:
            solver.max_step_size = mx.value
        if final is not None and self.primal is not None and self.done == self.max_iter:
            final.primal_var = self.primal
            final.primal_objective = torch.tensor(rows[-1][7], dtype=torch.float64, device=self.device).to(self.dtype)
        obj_log = rows[:, 0].tolist()
        return SolverResult(
            dual_val=dual_val,
            dual_objective=obj_log[-1] if obj_log else 0.0,
            objective_result=final,
            dual_objective_log=obj_log,
            step_size_log=rows[:, 1].tolist(),
        )
