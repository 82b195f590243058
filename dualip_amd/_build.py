"""Builds libdualip_hip.so (the C-ABI HIP library, include/dualip_hip.h) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdualip_hip.so")
SOURCES = ["api.hip", "matching_kernels.hip", "matching_kernels4.hip", "agd_kernels.hip", "lp_kernels.hip"]
HEADERS = ["common.h", "wave.h", "simplex.h", "simplex4.h", "fused_common.h", os.path.join("..", "..", "include", "dualip_hip.h")]
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-ffp-contract=off",       # keep the reference's mul-then-add rounding (no FMA contraction)
    "-munsafe-fp-atomics",     # hardware ds_add_f32/f64 and global_atomic_add for the gradient scatter
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libdualip_hip.so")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), *FLAGS, "-o", LIB_PATH, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
