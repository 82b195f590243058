"""The C ABI used the way INTEGRATION.md (option B) shows it: raw ``ctypes`` on ``libdualip_hip.so``, plain device pointers
taken from torch tensors, no import of the ``dualip_amd`` Python layer on the call path.  Checks one fused pass, the dual
epilogue and a device-resident AGD run against the oracle / the reference's golden trace."""
import ctypes
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import agd_oracle
from tests.helpers import load, problem, relerr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "dualip_amd", "lib", "libdualip_hip.so")
DL_F32, DL_F64, DL_I32, DL_I64 = 0, 1, 0, 1


class Proj(ctypes.Structure):  # dl_proj_desc
    _fields_ = [("kind", ctypes.c_int32), ("flags", ctypes.c_int32), ("p0", ctypes.c_double), ("p1", ctypes.c_double)]


def vp(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def test_fused_pass_epilogue_and_agd_through_raw_ctypes():
    assert os.path.exists(LIB), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(LIB)
    lib.dl_last_error_string.restype = ctypes.c_char_p
    lib.dl_agd_x.restype = ctypes.c_void_p
    z = load("g2_syn2000.npz")
    p = problem(z)
    m, n, nnz = p["m"], p["n"], len(p["a"])
    dev = "cuda:0"
    torch.cuda.set_device(0)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    colptr = torch.from_numpy(p["colptr"]).to(dev)
    rows = torch.from_numpy(p["rowidx"]).to(dev)
    a = torch.from_numpy(p["a"]).to(dev)
    c = torch.from_numpy(p["c"]).to(dev)
    b = torch.from_numpy(p["b"]).to(dev)
    descs = (Proj * 1)(Proj(4, 0, 1.0, 0.0))  # simplex z = 1 for every column (col_proj = NULL)
    h = ctypes.c_void_p()
    rc = lib.dl_matching_create(ctypes.byref(h), ctypes.c_int64(m), ctypes.c_int64(n), ctypes.c_int64(nnz), vp(colptr), vp(rows), DL_I64, vp(a), vp(c), DL_F64,
                                descs, ctypes.c_int32(1), None, stream)
    assert rc == 0, lib.dl_last_error_string().decode()
    try:
        # one pass + epilogue
        gamma = 0.02
        lam = torch.from_numpy(np.random.default_rng(3).uniform(0, 0.01, m)).to(dev)
        packed = torch.empty(m + 2, dtype=torch.float64, device=dev)
        x = torch.empty(nnz, dtype=torch.float64, device=dev)
        assert lib.dl_matching_calculate(h, vp(lam), ctypes.c_double(gamma), vp(packed), vp(x), stream) == 0
        grad = torch.empty(m, dtype=torch.float64, device=dev)
        scal = torch.empty(6, dtype=torch.float64, device=dev)
        assert lib.dl_dual_epilogue(ctypes.c_int64(m), DL_F64, vp(packed), vp(b), vp(lam), ctypes.c_double(gamma), vp(grad), vp(scal), stream) == 0
        ax, obj0, ssq, xo = oracle.matching_calculate(m, n, p["colptr"], p["rowidx"], p["a"], p["c"], lam.cpu().numpy(), gamma, [("simplex", {"z": 1.0})], dtype=np.float64)
        g_o, obj_o, reg_o, *_ = agd_oracle.epilogue(ax, obj0, ssq, lam.cpu().numpy(), p["b"], gamma, np.float64)
        assert relerr(x.cpu().numpy(), xo) < 1e-9 and relerr(grad.cpu().numpy(), g_o) < 1e-9
        assert relerr(scal.cpu().numpy()[:2], [obj_o, reg_o]) < 1e-9
        # device-resident AGD: the reference's golden 60-iteration trace
        iters = 60
        f32 = np.float32
        t = np.zeros(iters + 2, dtype=f32)
        for i in range(1, iters + 2):
            t[i] = f32((1.0 + np.sqrt(float(f32(f32(1.0) + f32(f32(4.0) * f32(t[i - 1] * t[i - 1])))))) / 2.0)
        beta = torch.from_numpy(((f32(1.0) - t[1 : iters + 1]) / t[2 : iters + 2]).astype(f32))  # host array of max_iter floats
        s = ctypes.c_void_p()
        lam0 = torch.zeros(m, dtype=torch.float64, device=dev)
        rc = lib.dl_agd_create(ctypes.byref(s), ctypes.c_int64(m), DL_F64, ctypes.c_int64(iters), vp(beta), ctypes.c_double(1e-3), ctypes.c_double(1e-1), None, vp(lam0), stream)
        assert rc == 0, lib.dl_last_error_string().decode()
        try:
            g_io = ctypes.c_double(gamma)
            rc = lib.dl_agd_run_matching(s, h, vp(b), ctypes.c_int64(1), ctypes.c_int64(iters), ctypes.byref(g_io), ctypes.c_int64(0), ctypes.c_double(1.0), None, stream)
            assert rc == 0, lib.dl_last_error_string().decode()
            rows_log = np.zeros((iters, 8), dtype=np.float64)
            assert lib.dl_agd_read_log(s, ctypes.c_int64(0), ctypes.c_int64(iters), ctypes.c_void_p(rows_log.ctypes.data), stream) == 0
            assert relerr(rows_log[:, 0], z["simplex1|f64|dual_obj_log"]) < 1e-6
            assert np.allclose(rows_log[:40, 1], z["simplex1|f64|step_log"][:40], rtol=1e-5)
        finally:
            lib.dl_agd_destroy(s)
    finally:
        lib.dl_matching_destroy(h)
