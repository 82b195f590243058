"""Builds libdualip_hip.so (the C-ABI HIP library, include/dualip_hip.h) in-tree with hipcc for gfx950."""
import fcntl
import hashlib
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdualip_hip.so")
SOURCES = ["api.hip", "matching_kernels.hip", "matching_kernels4.hip", "matching_kernels4_f64.hip", "matching_kernels4_lanes.hip", "matching_kernels4_lanes_f64.hip", "agd_kernels.hip", "lp_kernels.hip", "comm.hip", "sell_build.hip", "csc_ops.hip", "pack_build.hip"]
HEADERS = ["common.h", "wave.h", "simplex.h", "simplex4.h", "fused_common.h", "comm.h", "sell.h", "fused4_kernel.h", "agd_step.h", os.path.join("..", "..", "include", "dualip_hip.h")]
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


HASH_PATH = LIB_PATH + ".srchash"


def source_hash() -> str:
    """Digest of everything the binary is made from.  Staleness is decided by content, not by modification times: the tree
    is copied to other machines (where times may not survive) and several ranks may import the package at once."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for name in SOURCES + HEADERS:
        path = os.path.join(CSRC, name)
        h.update(name.encode())
        if os.path.exists(path):
            with open(path, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    try:
        with open(HASH_PATH) as fh:
            return fh.read().strip() != source_hash()
    except OSError:
        return True


def _object_hash(name: str) -> str:
    """Digest of one translation unit: its source, every header (any of them may be included) and the flags."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for part in [name] + HEADERS:
        path = os.path.join(CSRC, part)
        h.update(part.encode())
        if os.path.exists(path):
            with open(path, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:20]


_BLOCK = __import__("re").compile(r"^\.LBB\d+_\d+:")
_KERNEL = __import__("re").compile(r"^(_Z\w+):")


def _spill_defects(asm_path: str):
    """Occurrences, in one device assembly file, of a VGPR spill store/reload that sits between the label of a basic block and the
    `s_or_b64 exec, exec, ...` that re-enables the lanes which skipped the preceding branch.  SGPR spills (v_writelane, which
    ignores exec) may legitimately stand there; a VGPR spill executes under the branch's partial exec mask, so wavefronts that
    skipped the branch entirely store nothing and later reload whatever the scratch slot held.  hipcc 7.2.0 (clang 22.0.0git
    roc-7.2.0) emitted exactly that in one instantiation of matching_fused_kernel4 (wrong results, then memory faults)."""
    out, kernel = [], "?"
    with open(asm_path, errors="replace") as fh:
        lines = fh.read().split("\n")
    i, n = 0, len(lines)
    while i < n:
        t = lines[i].strip()
        m = _KERNEL.match(t)
        if m:
            kernel = m.group(1)
        if _BLOCK.match(t):
            j, spills = i + 1, []
            while j < n:
                u = lines[j].strip()
                if not u or u.startswith(";") or u.startswith(("v_writelane_b32", "v_readlane_b32", "s_nop", "s_waitcnt")):
                    j += 1
                elif u.startswith(("scratch_store", "scratch_load", "buffer_store", "buffer_load", "v_accvgpr_write", "v_accvgpr_read")) and ("Spill" in u or "Reload" in u):
                    spills.append(u.split(";")[0].strip())
                    j += 1
                else:
                    break
            if spills and j < n and lines[j].strip().startswith("s_or_b64 exec, exec,"):
                out += [f"{kernel} {t} {u}" for u in spills]
        i += 1
    return out


def _compile_objects(verbose: bool):
    """One object per source, compiled in parallel and kept by content hash: editing one file rebuilds one object."""
    from concurrent.futures import ThreadPoolExecutor

    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    cflags = [f for f in FLAGS if f != "-shared"]
    jobs, objs = [], []
    for name in SOURCES:
        obj = os.path.join(obj_dir, f"{os.path.splitext(name)[0]}.{_object_hash(name)}.o")
        objs.append(obj)
        if not os.path.exists(obj):
            jobs.append((name, obj))

    def one(job):
        name, obj = job
        # the object is assembled from device assembly kept beside it for a moment (--save-temps): that text is screened for a
        # code-generation defect of this compiler that once produced wrong results here (_spill_defects below)
        work = f"{obj}.{os.getpid()}.d"
        shutil.rmtree(work, ignore_errors=True)
        os.makedirs(work)
        tmp = os.path.join(work, "unit.o")
        cmd = [hipcc, *cflags, "-c", "--save-temps=obj", "-o", tmp, os.path.join(CSRC, name)]
        if verbose:
            print(" ".join(cmd).replace(tmp, obj))
        try:
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                return name, r.stdout + r.stderr
            found = []
            for f in sorted(os.listdir(work)):
                if f.endswith(".s") and "amdgcn" in f:
                    found += _spill_defects(os.path.join(work, f))
            if found and os.environ.get("DUALIP_BUILD_ALLOW_SPILL_DEFECT", "0") in ("", "0"):
                return name, ("hipcc placed a VGPR spill ahead of the exec restore of a control-flow join (lanes that skipped the branch never store "
                              "their value and reload garbage -- DESIGN.md section 8, tools/spill_exec_check.py):\n  " + "\n  ".join(found) +
                              "\nchange the register pressure of that kernel (or set DUALIP_BUILD_ALLOW_SPILL_DEFECT=1 to build anyway)")
            os.replace(tmp, obj)
            return name, None
        finally:
            shutil.rmtree(work, ignore_errors=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
            failed = [(n, err) for n, err in pool.map(one, jobs) if err is not None]
        if failed:
            raise RuntimeError("hipcc failed:\n" + "\n".join(f"--- {n}\n{err}" for n, err in failed))
    keep = set(objs)
    for f in os.listdir(obj_dir):  # objects of older source states
        full = os.path.join(obj_dir, f)
        if f.endswith(".o") and full not in keep:
            try:
                os.remove(full)
            except OSError:
                pass
    return hipcc, objs


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)  # one builder at a time; the others find a fresh binary when they get the lock
        try:
            if not force and not is_stale():
                return LIB_PATH
            if force:
                shutil.rmtree(os.path.join(LIB_DIR, "obj"), ignore_errors=True)
            hipcc, objs = _compile_objects(verbose)
            tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs, "-ldl"]
            if verbose:
                print(" ".join(cmd).replace(tmp, LIB_PATH))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc (link) failed:\n" + r.stdout + r.stderr)
            os.replace(tmp, LIB_PATH)  # atomic: a process that already mapped the old file keeps it
            with open(HASH_PATH + ".tmp", "w") as fh:
                fh.write(source_hash())
            os.replace(HASH_PATH + ".tmp", HASH_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
