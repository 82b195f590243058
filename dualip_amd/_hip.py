"""ctypes binding of libdualip_hip.so (include/dualip_hip.h).

The library is the product's only compute path: there is no Python/ATen fallback.  If the shared object cannot be
loaded (not built, hipcc missing) every entry point raises -- loudly -- instead of computing on the CPU.

``import torch`` happens before the library is opened so that ``libamdhip64.so.7`` resolves to the HIP runtime
PyTorch-ROCm already loaded (same streams, same allocations).
"""
import ctypes
from typing import Optional
import os

import torch  # noqa: F401  (must precede the CDLL open, see module docstring)

from . import _build

DL_F32, DL_F64 = 0, 1
DL_I32, DL_I64 = 0, 1
PROJ_NONE, PROJ_BOX, PROJ_CONE_LOWER, PROJ_CONE_UPPER, PROJ_SIMPLEX, PROJ_SIMPLEX_EQ = range(6)
LOG_COLS = 8
ABI_VERSION = 302  # dl_version(): bumped whenever an entry point's signature or a struct layout changes
PROJ_FLAG_BISECTION = 1
PROJ_FLAG_NO_SLICES = 2

_c_i64 = ctypes.c_int64
_c_vp = ctypes.c_void_p
_c_dbl = ctypes.c_double
_c_int = ctypes.c_int


class ProjDesc(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("flags", ctypes.c_int32), ("p0", ctypes.c_double), ("p1", ctypes.c_double)]


# name -> (restype, argtypes); mirrors include/dualip_hip.h one to one (tests check the export list against the header)
_SIGNATURES = {
    "dl_last_error_string": (ctypes.c_char_p, []),
    "dl_version": (_c_int, []),
    "dl_switch_name": (ctypes.c_char_p, [_c_int]),
    "dl_matching_create": (
        _c_int,
        [ctypes.POINTER(_c_vp), _c_i64, _c_i64, _c_i64, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_int, ctypes.POINTER(ProjDesc), ctypes.c_int32, _c_vp, _c_vp],
    ),
    "dl_matching_create2": (
        _c_int,
        [ctypes.POINTER(_c_vp), _c_i64, _c_i64, _c_i64, _c_vp, _c_int, _c_vp, _c_int, _c_vp, _c_vp, _c_int, ctypes.POINTER(ProjDesc), ctypes.c_int32, _c_vp, _c_vp],
    ),
    "dl_stage_to_device": (_c_int, [_c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int, ctypes.POINTER(_c_i64), ctypes.POINTER(ctypes.c_double)]),
    "dl_matching_destroy": (_c_int, [_c_vp]),
    "dl_matching_update_costs": (_c_int, [_c_vp, _c_vp]),
    "dl_matching_update_values": (_c_int, [_c_vp, _c_vp]),
    "dl_matching_set_fairness": (_c_int, [_c_vp, _c_vp, _c_vp]),
    "dl_matching_info": (_c_i64, [_c_vp, _c_int]),
    "dl_matching_calculate": (_c_int, [_c_vp, _c_vp, _c_dbl, _c_vp, _c_vp, _c_vp]),
    "dl_matching_profile": (_c_int, [_c_vp, _c_int]),
    "dl_matching_profile_read": (_c_int, [_c_vp, ctypes.POINTER(_c_dbl), ctypes.POINTER(_c_i64)]),
    "dl_matching_set_eq_padding": (_c_int, [_c_vp, _c_vp, ctypes.c_int32, _c_vp]),
    "dl_matching_timeline_read": (_c_int, [_c_vp, ctypes.POINTER(ctypes.c_uint64), _c_i64]),
    "dl_dual_epilogue": (_c_int, [_c_i64, _c_int, _c_vp, _c_vp, _c_vp, _c_dbl, _c_vp, _c_vp, _c_vp]),
    "dl_agd_create": (_c_int, [ctypes.POINTER(_c_vp), _c_i64, _c_int, _c_i64, _c_vp, _c_dbl, _c_dbl, _c_vp, _c_vp, _c_vp]),
    "dl_agd_destroy": (_c_int, [_c_vp]),
    "dl_agd_x": (_c_vp, [_c_vp]),
    "dl_agd_y": (_c_vp, [_c_vp]),
    "dl_agd_grad": (_c_vp, [_c_vp]),
    "dl_agd_get": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp]),
    "dl_agd_step": (_c_int, [_c_vp, _c_vp, _c_vp, _c_dbl, _c_i64, _c_int, _c_dbl, _c_vp]),
    "dl_agd_run_matching": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, ctypes.POINTER(_c_dbl), _c_i64, _c_dbl, _c_vp, _c_vp]),
    "dl_agd_run_matching_sharded": (_c_int, [_c_vp, ctypes.POINTER(_c_vp), ctypes.c_int32, _c_vp, _c_vp, _c_i64, _c_i64, ctypes.POINTER(_c_dbl), _c_i64, _c_dbl, _c_vp]),
    "dl_comm_rccl_unique_id": (_c_int, [_c_vp]),
    "dl_comm_create_rccl": (_c_int, [ctypes.POINTER(_c_vp), ctypes.c_int32, ctypes.c_int32, _c_vp, _c_i64]),
    "dl_comm_adopt_rccl": (_c_int, [ctypes.POINTER(_c_vp), _c_vp, _c_i64]),
    "dl_comm_p2p_begin": (_c_int, [ctypes.POINTER(_c_vp), ctypes.c_int32, ctypes.c_int32, _c_i64, _c_vp]),
    "dl_comm_p2p_connect": (_c_int, [_c_vp, _c_vp]),
    "dl_comm_destroy": (_c_int, [_c_vp]),
    "dl_comm_info": (_c_i64, [_c_vp, _c_int]),
    "dl_allreduce_sum": (_c_int, [_c_vp, _c_vp, _c_i64, _c_vp]),
    "dl_comm_check": (_c_int, [_c_vp, _c_vp]),
    "dl_matching_own_inputs": (_c_int, [_c_vp, _c_vp]),
    "dl_comm_status": (_c_int, [_c_vp, ctypes.POINTER(ctypes.c_int32), _c_vp]),
    "dl_comm_inject_fault": (_c_int, [_c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64]),
    "dl_comm_set_emulation": (_c_int, [_c_vp, _c_dbl]),
    "dl_comm_set_fenced": (_c_int, [_c_vp, _c_int]),
    "dl_comm_selftest": (_c_int, [_c_vp, ctypes.c_int32, ctypes.c_uint64, _c_i64, ctypes.POINTER(_c_i64), _c_vp]),
    "dl_comm_reset": (_c_int, [_c_vp, _c_vp]),
    "dl_comm_set_timeout_ms": (_c_int, [_c_vp, _c_i64]),
    "dl_comm_profile": (_c_int, [_c_vp, _c_int]),
    "dl_comm_profile_read": (_c_int, [_c_vp, ctypes.POINTER(_c_dbl), ctypes.POINTER(_c_i64)]),
    "dl_agd_read_log": (_c_int, [_c_vp, _c_i64, _c_i64, _c_vp, _c_vp]),
    "dl_agd_read_max_step": (_c_int, [_c_vp, ctypes.POINTER(_c_dbl), _c_vp]),
    "dl_project_dense": (_c_int, [_c_i64, _c_i64, _c_int, _c_vp, _c_vp, ctypes.POINTER(ProjDesc), _c_vp]),
    "dl_measure_read_bandwidth": (_c_int, [_c_vp, _c_i64, ctypes.c_int32, ctypes.POINTER(_c_dbl), _c_vp]),
    "dl_csc_scale_rows": (_c_int, [_c_i64, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp]),
    "dl_csc_scale_cols": (_c_int, [_c_i64, _c_i64, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp]),
    "dl_csc_elementwise": (_c_int, [_c_i64, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_vp]),
    "dl_csc_row_sums": (_c_int, [_c_i64, _c_i64, _c_vp, _c_int, _c_vp, _c_vp, _c_int, _c_vp]),
    "dl_csc_project_columns": (_c_int, [_c_i64, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, ctypes.POINTER(ProjDesc), _c_int, _c_vp]),
    "dl_jacobi_precondition": (_c_int, [_c_i64, _c_i64, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp]),
    "dl_lp_create": (_c_int, [ctypes.POINTER(_c_vp), _c_i64, _c_i64, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_int]),
    "dl_lp_destroy": (_c_int, [_c_vp]),
    "dl_lp_calculate": (_c_int, [_c_vp, _c_vp, _c_dbl, _c_vp, _c_vp, _c_vp]),
    "dl_lp_primal": (_c_int, [_c_vp, _c_vp, _c_dbl, _c_int, _c_vp, _c_vp]),
    "dl_lp_gradient": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib_path() -> str:
    return _build.LIB_PATH


def load():
    """Open libdualip_hip.so (building it with hipcc when the in-tree binary is missing or stale)."""
    global _lib
    if _lib is not None:
        return _lib
    stale = None
    if os.environ.get("DUALIP_DEV_LIBRARY", "0") not in ("", "0"):
        # tools/ only: the developer build, whose ablation switches skip work inside the kernels (wrong results on purpose)
        import warnings

        warnings.warn("DUALIP_DEV_LIBRARY=1: loading libdualip_hip_dev.so (developer build: DUALIP_HIP_ABLATE and the tuning switches are live)", RuntimeWarning)
        handle = ctypes.CDLL(_build.build_dev())
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
        return _lib
    try:
        path = _build.build()
    except Exception as exc:  # no silent fallback: the HIP path is the product
        if os.path.exists(_build.LIB_PATH):
            path, stale = _build.LIB_PATH, exc  # present but not rebuilt (e.g. hipcc unavailable on this box)
        else:
            raise HipLibraryError(f"libdualip_hip.so is not built and could not be built: {exc}") from exc
    try:
        handle = ctypes.CDLL(path)
    except OSError as exc:
        raise HipLibraryError(f"cannot load {path}: {exc}") from exc
    if stale is not None:
        # a binary whose sources have changed since it was built may not have the ABI _SIGNATURES describes: only accept it
        # when it reports the ABI version these bindings were written for and exports every entry point, and say so
        import warnings

        missing = [n for n in _SIGNATURES if not hasattr(handle, n)]
        handle.dl_version.restype = _c_int
        if missing or int(handle.dl_version()) != ABI_VERSION:
            raise HipLibraryError(f"{path} is stale (rebuild failed: {stale}) and does not match these bindings "
                                  f"(ABI {int(handle.dl_version())} vs {ABI_VERSION}; missing {missing[:5]})") from stale
        warnings.warn(f"libdualip_hip.so could not be rebuilt ({stale}); using the existing binary, whose ABI version matches", RuntimeWarning)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    if int(handle.dl_version()) != ABI_VERSION:
        raise HipLibraryError(f"{path} reports ABI version {int(handle.dl_version())}, these bindings expect {ABI_VERSION}")
    _lib = handle
    return _lib


def last_error() -> str:
    return load().dl_last_error_string().decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map a C-ABI status to the exception type the reference raises for the same mistake."""
    if rc == 0:
        return
    msg = load().dl_last_error_string().decode("utf-8", "replace")
    if rc in (1, 2, 3):
        raise ValueError(msg)
    if rc == 5:
        raise MemoryError(msg)
    raise RuntimeError(f"libdualip_hip status {rc}: {msg}")


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return DL_F32
    if dtype == torch.float64:
        return DL_F64
    raise ValueError(f"dualip_amd supports float32 and float64 values, got {dtype}")


def idx_code(dtype: torch.dtype) -> int:
    if dtype == torch.int32:
        return DL_I32
    if dtype == torch.int64:
        return DL_I64
    raise ValueError(f"CSC indices must be int32 or int64, got {dtype}")


def require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise HipLibraryError(
            f"{what} lives on '{t.device}': dualip_amd computes on an AMD GPU through libdualip_hip.so only "
            "(there is no CPU fallback); move the inputs to a ROCm device"
        )


_stage_logged = False


def compute_device() -> torch.device:
    """The ROCm device CPU-resident inputs are staged to: the process's current one (one process per GPU sets it per rank)."""
    if not torch.cuda.is_available():
        raise HipLibraryError(
            "dualip_amd computes on an AMD GPU through libdualip_hip.so only (there is no CPU fallback) and this process sees no ROCm device: "
            "host_device='cpu' inputs are staged to a GPU, not computed on the host"
        )
    return torch.device("cuda", torch.cuda.current_device())


def note_staging(what: str, src_device, device=None):
    """The once-per-process log line of ``stage`` (and its "no GPU" error) WITHOUT copying anything: for callers that move a whole input
    bundle themselves right afterwards (a ``stage(values)`` whose result is thrown away costs a second host-to-device copy of every value)."""
    global _stage_logged
    dev = compute_device() if device is None else device
    if not _stage_logged:
        _stage_logged = True
        import logging

        logging.getLogger("dualip_amd").warning("%s lives on '%s': staging CPU inputs to %s for the HIP path and returning results on the caller's device "
                                                "(no CPU compute path exists; said once)", what, src_device, dev)
    return dev


def stage(t, what: str, device=None):
    """``t`` on a ROCm device.  The reference's callers default to ``host_device="cpu"`` (examples/movielens_matching/
    movies_lens_matching.py:227, examples/miplib_2017/solve_miplib_dataset.py:58) and its tests build CPU tensors: such inputs are COPIED
    to the current ROCm device (said once in the log) and the results handed back on the caller's device -- the arithmetic still runs in
    libdualip_hip.so only.  Without a GPU this raises, like every other entry point."""
    global _stage_logged
    if t is None or t.is_cuda:
        return t
    return t.to(note_staging(what, t.device, device))


DL_U16 = 2  # dl_matching_create2: row indices narrowed to 16 bits on their way to the device
STAGING_LOG = []  # one record per stage_array call: {"what", "bytes_host", "bytes_link", "seconds"} (tools/host_buffers_rate.py reads it)


def stage_array(t: torch.Tensor, device, narrow_to: Optional[torch.dtype] = None, what: str = "array") -> torch.Tensor:
    """A contiguous CPU tensor (ordinary pageable memory) -> a new device tensor through ``dl_stage_to_device``: host threads fill pinned
    16 MB buffers and queue their DMAs back to back (include/dualip_hip.h).  ``narrow_to`` (int64/int32 input only): torch.int32, or
    torch.uint16 (returned as an int16-typed tensor holding the 16-bit patterns -- torch has no arithmetic on uint16; only the C library reads
    it): the integers are narrowed ON THE HOST, so an int64 index array crosses the link at half / a quarter of its size; a value that does
    not fit raises ValueError."""
    import time as _time

    if t.is_cuda:
        raise ValueError("stage_array takes a host tensor")
    src = t.contiguous()
    lib = load()
    dev = torch.device(device)
    if narrow_to is None:
        out = torch.empty(src.shape, dtype=src.dtype, device=dev)
        sb = db = src.element_size()
        unsigned = 0
    else:
        if src.dtype not in (torch.int64, torch.int32):
            raise ValueError("only integer tensors are narrowed")
        out_dtype, db, unsigned = {torch.int32: (torch.int32, 4, 0), torch.uint16: (torch.int16, 2, 1)}[narrow_to]
        sb = src.element_size()
        if db >= sb:
            raise ValueError(f"cannot narrow {src.dtype} to {narrow_to}")
        out = torch.empty(src.shape, dtype=out_dtype, device=dev)
    bad, secs = ctypes.c_int64(0), ctypes.c_double(0.0)
    t0 = _time.perf_counter()
    with torch.cuda.device(dev):
        torch.cuda.current_stream(dev).synchronize()  # (`out` was allocated on torch's stream: nothing of an earlier owner of that memory may still be running)
        check(lib.dl_stage_to_device(ptr(out), src.data_ptr(), src.numel(), sb, db, unsigned, int(os.environ.get("DUALIP_STAGE_THREADS", "0") or 0), ctypes.byref(bad), ctypes.byref(secs)))
    if bad.value:
        raise ValueError(f"{what}: {bad.value} value(s) do not fit {narrow_to}")
    STAGING_LOG.append({"what": what, "bytes_host": src.numel() * sb, "bytes_link": src.numel() * db, "seconds": _time.perf_counter() - t0})
    return out


def result_to(res, device):
    """An ObjectiveResult / SolverResult whose tensors live on ``device`` (the caller's, when its inputs were staged)."""
    import dataclasses

    if res is None or device is None:
        return res
    moved = {}
    for f in dataclasses.fields(res):
        v = getattr(res, f.name)
        if isinstance(v, torch.Tensor):
            moved[f.name] = v.to(device)
        elif dataclasses.is_dataclass(v) and not isinstance(v, type):
            moved[f.name] = result_to(v, device)
    return dataclasses.replace(res, **moved) if moved else res


def stream_ptr(device=None) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def ptr(t) -> int:
    return 0 if t is None else int(t.data_ptr())
