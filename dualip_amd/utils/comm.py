"""Communicator of the column-sharded objective: the Python face of ``dl_comm`` (include/dualip_hip.h).

The reference exchanges through ``torch.distributed`` -- three ``reduce`` calls, a ``barrier`` and two ``broadcast`` calls per
iteration (src/dualip/objectives/matching.py:272-277, src/dualip/optimizers/agd.py:204-206).  Here ``torch.distributed`` is
only the side channel that sets a communicator up (RCCL unique id / hipIpc handles travel through ``all_gather_object``);
the per-iteration exchange itself is issued by the C library, inside its device-resident loop.

Back-ends (``DUALIP_COMM`` = ``auto`` | ``p2p`` | ``rccl``):
  * ``p2p``   one-shot all-to-all over hipIpc-mapped mailboxes fused into the slab-reduction / step kernels;
  * ``rccl``  ``ncclAllReduce`` on a communicator the library creates for itself;
  * ``auto``  p2p when it passes a self-test on every rank (known sums through both mailbox parities), else rccl; ranks that
              share a device (single-GPU test harness) can only use p2p.
"""
import ctypes
import os
from typing import Optional

import torch
import torch.distributed as dist

from dualip_amd import _hip

RCCL, P2P = 1, 2
_NAMES = {RCCL: "rccl", P2P: "p2p"}


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _gather(obj, group, world):
    if world == 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj, group=group)
    return out


class Communicator:
    """Sum-all-reduce of ``count`` doubles among the ranks of ``group`` (default group when None), device ``device``."""

    def __init__(self, count: int, device, group=None, backend: Optional[str] = None):
        self.lib = _hip.load()
        self.device = torch.device(device)
        self.group = group
        self.count = int(count)
        self.world, self.rank = _world(group)
        self.handle = ctypes.c_void_p()
        want = (backend or os.environ.get("DUALIP_COMM", "auto")).lower()
        if want not in ("auto", "p2p", "rccl"):
            raise ValueError(f"DUALIP_COMM must be auto, p2p or rccl, got {want}")
        # ranks sharing a device cannot form an RCCL communicator
        props = torch.cuda.get_device_properties(self.device)
        ident = (os.uname().nodename, getattr(props, "pci_bus_id", None), os.environ.get("HIP_VISIBLE_DEVICES", ""),
                 os.environ.get("CUDA_VISIBLE_DEVICES", ""), self.device.index)
        idents = _gather(ident, group, self.world)
        shared_device = len(set(idents)) < len(idents)
        one_node = len({i[0] for i in idents}) == 1
        self.fallback_reason = None
        if want == "rccl" and shared_device and self.world > 1:
            raise RuntimeError("DUALIP_COMM=rccl: RCCL cannot place two ranks on one device")
        if want in ("auto", "p2p") and one_node and self.world <= 16:
            ok, why = self._try_p2p()
            oks = _gather((ok, why), group, self.world)
            if all(o for o, _ in oks):
                return
            self._destroy()
            self.fallback_reason = "; ".join(sorted({w for o, w in oks if not o and w}))
            if want == "p2p" or shared_device:
                raise RuntimeError(f"P2P exchange unavailable: {self.fallback_reason}")
        elif want == "p2p":
            raise RuntimeError("DUALIP_COMM=p2p needs all ranks on one node (at most 16)")
        self._make_rccl()

    # ---- construction ---------------------------------------------------------------------------------------
    def _try_p2p(self):
        lib = self.lib
        hd = (ctypes.c_ubyte * 64)()
        with torch.cuda.device(self.device):
            rc = lib.dl_comm_p2p_begin(ctypes.byref(self.handle), self.world, self.rank, self.count, hd)
        mine = bytes(hd) if rc == 0 else None
        why = None if rc == 0 else _hip.last_error()
        handles = _gather(mine, self.group, self.world)
        if any(h is None for h in handles):
            return False, why or "another rank could not allocate its mailbox"
        blob = (ctypes.c_ubyte * (64 * self.world)).from_buffer_copy(b"".join(handles))
        with torch.cuda.device(self.device):
            rc = lib.dl_comm_p2p_connect(self.handle, blob)
        if rc != 0:
            why = _hip.last_error()
        oks = _gather(rc == 0, self.group, self.world)  # (also the barrier: every rank has mapped every mailbox)
        if not all(oks):
            return False, why or "another rank could not map the mailboxes"
        return self._self_test()

    def _self_test(self):
        """Known sums through both mailbox parities, three rounds: every element must come back exact."""
        n = min(self.count, 4096)
        base = torch.arange(1, n + 1, dtype=torch.float64, device=self.device)
        want = base * (self.world * (self.world + 1) / 2.0)
        # a mapping that does not carry remote stores into a running kernel shows as a timed-out wait: keep that short here
        user_ms = int(os.environ.get("DUALIP_COMM_TIMEOUT_MS", "0") or 0)
        self.lib.dl_comm_set_timeout_ms(self.handle, min(user_ms, 2000) if user_ms > 0 else 2000)
        try:
            for rnd in range(4):
                v = base * float(self.rank + 1)
                self.all_reduce_(v)
                self.check()
                if not torch.equal(v, want):
                    return False, f"P2P self-test round {rnd}: wrong sums on rank {self.rank}"
        except Exception as exc:  # timeouts surface here
            return False, f"P2P self-test failed on rank {self.rank}: {exc}"
        finally:
            if self.handle is not None and self.handle.value:
                self.lib.dl_comm_set_timeout_ms(self.handle, user_ms if user_ms > 0 else 20000)
        return True, None

    def _make_rccl(self):
        """Every rank raises, or none does: a rank that cannot load RCCL must not leave the others inside ncclCommInitRank."""
        lib = self.lib
        uid = (ctypes.c_ubyte * 128)()
        rc = lib.dl_comm_rccl_unique_id(uid)  # (every rank: also the check that the RCCL entry points resolve here)
        why = None if rc == 0 else _hip.last_error()
        ids = _gather((bytes(uid), why), self.group, self.world)
        bad = sorted({w for _, w in ids if w})
        if bad:
            raise RuntimeError("RCCL unavailable: " + "; ".join(bad))
        blob = (ctypes.c_ubyte * 128).from_buffer_copy(ids[0][0])
        with torch.cuda.device(self.device):
            rc = lib.dl_comm_create_rccl(ctypes.byref(self.handle), self.world, self.rank, blob, self.count)
        why = None if rc == 0 else _hip.last_error()
        bad = sorted({w for w in _gather(why, self.group, self.world) if w})
        if bad:
            self._destroy()
            raise RuntimeError("RCCL communicator could not be created: " + "; ".join(bad))

    # ---- use ------------------------------------------------------------------------------------------------
    @property
    def backend(self) -> str:
        return _NAMES.get(int(self.lib.dl_comm_info(self.handle, 0)), "?")

    @property
    def exchanges(self) -> int:
        return int(self.lib.dl_comm_info(self.handle, 3))

    def info(self) -> dict:
        return {"backend": self.backend, "world": int(self.lib.dl_comm_info(self.handle, 1)), "rank": int(self.lib.dl_comm_info(self.handle, 2)),
                "fallback_reason": self.fallback_reason}

    def all_reduce_(self, buf: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks of a float64 device vector (asynchronous on the current stream)."""
        if buf.dtype != torch.float64 or not buf.is_cuda or not buf.is_contiguous():
            raise ValueError("all_reduce_ takes a contiguous float64 device tensor")
        with torch.cuda.device(self.device):
            _hip.check(self.lib.dl_allreduce_sum(self.handle, _hip.ptr(buf), buf.numel(), _hip.stream_ptr(self.device)))
        return buf

    def rendezvous(self) -> None:
        """Host-side meeting of the ranks (through the process group, not the device): called once before a device-resident
        solve so that the first in-kernel wait of the P2P exchange starts with the ranks milliseconds, not a handle creation,
        apart -- the in-kernel waits are bounded (20 s, DUALIP_COMM_TIMEOUT_MS)."""
        _gather(0, self.group, self.world)

    def check(self) -> None:
        """Synchronise and raise if a P2P wait ever timed out."""
        with torch.cuda.device(self.device):
            _hip.check(self.lib.dl_comm_check(self.handle, _hip.stream_ptr(self.device)))

    def set_emulation(self, scale: float) -> None:
        _hip.check(self.lib.dl_comm_set_emulation(self.handle, float(scale)))

    def profile(self, enable) -> None:
        _hip.check(self.lib.dl_comm_profile(self.handle, int(enable)))

    def profile_read(self):
        ms, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
        _hip.check(self.lib.dl_comm_profile_read(self.handle, ctypes.byref(ms), ctypes.byref(cnt)))
        return int(cnt.value), float(ms.value)

    def _destroy(self):
        if self.handle is not None and self.handle.value:
            self.lib.dl_comm_destroy(self.handle)
        self.handle = ctypes.c_void_p()

    def close(self):
        self._destroy()

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass


def make_communicator(count: int, device, group=None, backend: Optional[str] = None):
    """``(Communicator, None)``, or ``(None, reason)`` when neither back-end can be set up on every rank and none was asked
    for by name: the caller then exchanges through ``torch.distributed`` (one all_reduce per iteration, issued from Python).
    A collective call; the outcome is the same on every rank."""
    import warnings

    explicit = (backend or os.environ.get("DUALIP_COMM", "auto")).lower() != "auto"
    comm, err = None, None
    try:
        if os.environ.get("DUALIP_COMM_DISABLE", "0") not in ("", "0"):  # (switch, and what the fallback's test sets)
            raise RuntimeError("native exchange switched off by DUALIP_COMM_DISABLE")
        comm = Communicator(count, device, group=group, backend=backend)
    except Exception as exc:  # (Communicator raises on every rank or on none; the gather below covers what is left)
        err = f"{type(exc).__name__}: {exc}"
    world, rank = _world(group)
    errs = sorted({e for e in _gather(err, group, world) if e})
    if not errs:
        return comm, None
    if comm is not None:
        comm.close()
    reason = "; ".join(errs)
    if explicit:
        raise RuntimeError(reason)
    if rank == 0:
        warnings.warn(f"dualip_amd: no native exchange ({reason}); falling back to torch.distributed all_reduce from Python")
    return None, reason

