"""Communicator of the column-sharded objective: the Python face of ``dl_comm`` (include/dualip_hip.h).

The reference exchanges through ``torch.distributed`` -- three ``reduce`` calls, a ``barrier`` and two ``broadcast`` calls per
iteration (src/dualip/objectives/matching.py:272-277, src/dualip/optimizers/agd.py:204-206).  Here ``torch.distributed`` is
only the side channel that sets a communicator up (RCCL unique id / hipIpc handles travel through ``all_gather_object``);
the per-iteration exchange itself is issued by the C library, inside its device-resident loop.

Back-ends (``DUALIP_COMM`` = ``auto`` | ``p2p`` | ``p2p-fenced`` | ``rccl``):
  * ``p2p``         one-shot all-to-all over hipIpc-mapped mailboxes fused into the slab-reduction / step kernels, ordering by
                    write-through stores + drained vmcnt (no cache write-back / invalidate; dualip_amd/csrc/comm.h);
  * ``p2p-fenced``  the same exchange with the system-scope release / acquire fences the HIP memory model asks for (slower;
                    also ``DUALIP_COMM_FENCED=1``);
  * ``rccl``        ``ncclAllReduce`` on a communicator the library creates for itself;
  * ``auto``        three levels, decided collectively at creation: p2p when its soak test passes on EVERY rank
                    (``DUALIP_COMM_SOAK_ROUNDS``, default 1000 exchanges of m + 2 known values with randomised per-rank delays),
                    else p2p-fenced when ITS soak test passes, else rccl; ranks that share a device (single-GPU test harness)
                    can only use the p2p variants.  The caller falls back to ``torch.distributed`` when none can be set up
                    (``make_communicator``).  ``Communicator.backend`` / ``info()`` say which level a run landed on.
"""
import ctypes
import os
from typing import Optional

import torch
import torch.distributed as dist

from dualip_amd import _hip

RCCL, P2P = 1, 2
_NAMES = {RCCL: "rccl", P2P: "p2p"}
HEALTHY, TIMED_OUT, CHECKSUM = 0, 1, 2  # dl_comm_status
_STATUS = {TIMED_OUT: "a wait for another rank's partial sums timed out", CHECKSUM: "a slot's payload did not match the checksum its sender announced"}


class ExchangeError(RuntimeError):
    """The P2P exchange failed on some rank (raised on EVERY rank by ``Communicator.meet``): ``codes`` = dl_comm_status per rank."""

    def __init__(self, codes, backend):
        self.codes, self.backend = list(codes), backend
        bad = ", ".join(f"rank {r}: {_STATUS.get(c, c)}" for r, c in enumerate(self.codes) if c)
        hint = "" if backend != "p2p" else " (DUALIP_COMM=p2p-fenced or rccl orders the exchange by the book)"
        super().__init__(f"the {backend} exchange failed -- {bad}: the results of this run are invalid on every rank{hint}")


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
        if want not in ("auto", "p2p", "p2p-fenced", "rccl"):
            raise ValueError(f"DUALIP_COMM must be auto, p2p, p2p-fenced or rccl, got {want}")
        force_fenced = want == "p2p-fenced" or os.environ.get("DUALIP_COMM_FENCED", "0") not in ("", "0")
        # ranks sharing a device cannot form an RCCL communicator
        props = torch.cuda.get_device_properties(self.device)
        ident = (os.uname().nodename, getattr(props, "pci_bus_id", None), os.environ.get("HIP_VISIBLE_DEVICES", ""),
                 os.environ.get("CUDA_VISIBLE_DEVICES", ""), self.device.index)
        idents = _gather(ident, group, self.world)
        shared_device = len(set(idents)) < len(idents)
        self.shared_device = shared_device
        one_node = len({i[0] for i in idents}) == 1
        self.fallback_reason = None
        self.selftest = None  # what the creation-time soak test saw, per level tried
        self.auto = want == "auto" and not force_fenced  # free to move to a stricter level at run time (degrade)
        self.degraded = None  # set when a run-time checksum mismatch moved an `auto` communicator from p2p to p2p-fenced
        self.devices = [f"{i[0]}:{i[1] or i[4]}" for i in idents]  # every rank's node:PCI bus id (device ordinal when torch does not say)
        if want == "rccl" and shared_device and self.world > 1:
            raise RuntimeError("DUALIP_COMM=rccl: RCCL cannot place two ranks on one device")
        if want in ("auto", "p2p", "p2p-fenced") and one_node and self.world <= 16:
            ok, why = self._try_p2p(force_fenced)
            if ok:
                return
            self._destroy()
            self.fallback_reason = why
            if want in ("p2p", "p2p-fenced") or shared_device:
                raise RuntimeError(f"P2P exchange unavailable: {self.fallback_reason}")
        elif want in ("p2p", "p2p-fenced"):
            raise RuntimeError(f"DUALIP_COMM={want} needs all ranks on one node (at most 16)")
        self._make_rccl()

    # ---- construction ---------------------------------------------------------------------------------------
    def _try_p2p(self, force_fenced: bool = False):
        """(ok, reason) -- the same on every rank.  Maps the mailboxes, then soak-tests the default ordering and, if any rank
        saw a wrong element or a timed-out wait, the fenced one."""
        lib = self.lib
        hd = (ctypes.c_ubyte * 64)()
        with torch.cuda.device(self.device):
            rc = lib.dl_comm_p2p_begin(ctypes.byref(self.handle), self.world, self.rank, self.count, hd)
        mine = bytes(hd) if rc == 0 else None
        why = None if rc == 0 else _hip.last_error()
        handles = _gather((mine, why), self.group, self.world)
        if any(h is None for h, _ in handles):
            return False, "; ".join(sorted({w for _, w in handles if w})) or "a rank could not allocate its mailbox"
        blob = (ctypes.c_ubyte * (64 * self.world)).from_buffer_copy(b"".join(h for h, _ in handles))
        with torch.cuda.device(self.device):
            rc = lib.dl_comm_p2p_connect(self.handle, blob)
        why = None if rc == 0 else _hip.last_error()
        oks = _gather(why, self.group, self.world)  # (also the barrier: every rank has mapped every mailbox)
        if any(oks):
            return False, "; ".join(sorted({w for w in oks if w}))
        self.selftest = []
        reasons = []
        for fenced in ([True] if force_fenced else [False, True]):
            _hip.check(lib.dl_comm_set_fenced(self.handle, int(fenced)))
            ok, why = self._self_test()
            results = _gather((ok, why), self.group, self.world)  # (every rank has left the test)
            all_ok = all(o for o, _ in results)
            self.selftest.append({"variant": "p2p-fenced" if fenced else "p2p", "ok": all_ok, "rounds": self._soak_rounds(),
                                  "failures": sorted({w for o, w in results if not o and w})})
            if all_ok:
                return True, None
            reasons += [w for o, w in results if not o and w]
            with torch.cuda.device(self.device):
                _hip.check(lib.dl_comm_reset(self.handle, _hip.stream_ptr(self.device)))
            _gather(0, self.group, self.world)  # every rank is reset before anyone pushes again
        return False, "; ".join(sorted(set(reasons)))

    @staticmethod
    def _soak_rounds() -> int:
        return max(4, int(os.environ.get("DUALIP_COMM_SOAK_ROUNDS", "1000") or 1000))

    def _self_test(self):
        """Soak test of the exchange at creation (dl_comm_selftest): >= 1000 sum-all-reduces of m + 2 known values through both
        mailbox parities, every rank delayed by its own pseudo-random time before each push (ranks arrive in every order; a
        third of the rounds run back to back); every element must come back exact and no wait may time out."""
        # a mapping that does not carry remote stores into a running kernel shows as a timed-out wait: keep that short here
        # (ranks that SHARE a device -- the one-GPU test harness -- are time-sliced against each other by the driver: a rank's in-kernel wait then
        #  spans the other processes' turns; eight of them beside a test runner holding its own context overran the two seconds, and the ranks
        #  that gave up raced ahead of the others' readers.  There the regular bound applies: the mapping is to the device's own memory.)
        user_ms = int(os.environ.get("DUALIP_COMM_TIMEOUT_MS", "0") or 0)
        soak_ms = (user_ms if user_ms > 0 else 20000) if getattr(self, "shared_device", False) else (min(user_ms, 2000) if user_ms > 0 else 2000)
        self.lib.dl_comm_set_timeout_ms(self.handle, soak_ms)
        bad = ctypes.c_int64(-1)
        variant = "p2p-fenced" if int(self.lib.dl_comm_info(self.handle, 5)) == 1 else "p2p"
        try:
            if variant in os.environ.get("DUALIP_COMM_TEST_FAIL", "").split(","):  # test hook: pretend this level failed here
                raise RuntimeError(f"DUALIP_COMM_TEST_FAIL names {variant}")
            with torch.cuda.device(self.device):
                _hip.check(self.lib.dl_comm_selftest(self.handle, self._soak_rounds(), 20250929, int(os.environ.get("DUALIP_COMM_SOAK_DELAY_US", "40") or 0),
                                                     ctypes.byref(bad), _hip.stream_ptr(self.device)))
            self.check()
            if bad.value != 0:
                return False, f"P2P soak test: {bad.value} wrong elements on rank {self.rank}"
        except Exception as exc:  # timeouts surface here
            return False, f"P2P soak test failed on rank {self.rank}: {exc}"
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
        name = _NAMES.get(int(self.lib.dl_comm_info(self.handle, 0)), "?")
        return "p2p-fenced" if (name == "p2p" and int(self.lib.dl_comm_info(self.handle, 5)) == 1) else name

    @property
    def exchanges(self) -> int:
        return int(self.lib.dl_comm_info(self.handle, 3))

    def info(self) -> dict:
        """What a run can print to show which exchange it really used: back-end, world and rank as the C handle holds them, the rank
        count and rank the RCCL communicator ITSELF reports (None for P2P), every rank's device, the creation-time soak test."""
        rccl_n, rccl_r = int(self.lib.dl_comm_info(self.handle, 7)), int(self.lib.dl_comm_info(self.handle, 8))
        return {"backend": self.backend, "world": int(self.lib.dl_comm_info(self.handle, 1)), "rank": int(self.lib.dl_comm_info(self.handle, 2)),
                "rccl_reported_world": rccl_n if rccl_n >= 0 else None, "rccl_reported_rank": rccl_r if rccl_r >= 0 else None,
                "devices": self.devices, "distinct_devices": len(set(self.devices)), "device_ordinal": int(self.lib.dl_comm_info(self.handle, 6)),
                "payload_checksums": self.backend.startswith("p2p"), "degraded": self.degraded,
                "fallback_reason": self.fallback_reason, "creation_selftest": self.selftest}

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
        """Synchronise and raise if a P2P exchange ever failed ON THIS RANK (a timed-out wait, a payload checksum mismatch).
        Rank-local: inside a solve use ``meet`` so that every rank stops together."""
        with torch.cuda.device(self.device):
            _hip.check(self.lib.dl_comm_check(self.handle, _hip.stream_ptr(self.device)))

    def status(self) -> int:
        """Synchronise; 0 healthy, 1 a wait timed out, 2 payload checksum mismatch (sticky, this rank)."""
        code = ctypes.c_int32(0)
        with torch.cuda.device(self.device):
            _hip.check(self.lib.dl_comm_status(self.handle, ctypes.byref(code), _hip.stream_ptr(self.device)))
        return int(code.value)

    def meet(self) -> None:
        """Host-side meeting of the ranks that also exchanges their health (a collective call): if the exchange failed on ANY
        rank, EVERY rank raises ``ExchangeError`` here -- a rank that saw a failure does not throw alone while its peers block in
        the next collective until the process group times out."""
        codes = _gather(self.status(), self.group, self.world)
        if any(codes):
            raise ExchangeError(codes, self.backend)

    def degrade(self) -> bool:
        """After ``meet`` raised for a checksum mismatch: move an ``auto`` communicator from the unfenced P2P ordering to the
        fenced one (collective; the same decision on every rank) and clear the error state, so that the caller can repeat its run.
        False when there is no stricter level to move to (explicitly chosen back-end, already fenced, RCCL)."""
        can = self.auto and self.backend == "p2p"
        if not all(_gather(can, self.group, self.world)):
            return False
        with torch.cuda.device(self.device):
            _hip.check(self.lib.dl_comm_reset(self.handle, _hip.stream_ptr(self.device)))
        _hip.check(self.lib.dl_comm_set_fenced(self.handle, 1))
        _gather(0, self.group, self.world)  # every rank is reset and fenced before anyone pushes again
        self.degraded = "p2p -> p2p-fenced at run time (a payload checksum did not match under the unfenced ordering)"
        return True

    def inject_fault(self, kind: int, target_rank: int, at_exchange: Optional[int] = None) -> None:
        """Test hook: damage this rank's contribution to exchange ``at_exchange`` (default: the next) as stored into
        ``target_rank``'s mailbox -- 1 = a flipped bit in one element, 2 = data stores dropped (stale slot), 0 = disarm."""
        seq = self.exchanges + 1 if at_exchange is None else int(at_exchange)
        _hip.check(self.lib.dl_comm_inject_fault(self.handle, int(kind), int(target_rank), seq))

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

