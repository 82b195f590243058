"""Builds libdualip_hip.so (the C-ABI HIP library, include/dualip_hip.h) in-tree with hipcc for gfx950."""
import fcntl
import hashlib
import json
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdualip_hip.so")
SOURCES = ["api.hip", "matching_kernels.hip", "matching_kernels4.hip", "matching_kernels4_f64.hip", "matching_kernels4_lanes.hip", "matching_kernels4_lanes_f64.hip", "agd_kernels.hip", "lp_kernels.hip", "comm.hip", "sell_build.hip", "csc_ops.hip", "pack_build.hip", "stage.hip"]
HEADERS = ["common.h", "wave.h", "simplex.h", "simplex4.h", "fused_common.h", "comm.h", "sell.h", "fused4_kernel.h", "agd_step.h", os.path.join("..", "..", "include", "dualip_hip.h")]
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-ffp-contract=off",       # keep the reference's mul-then-add rounding (no FMA contraction)
    "-munsafe-fp-atomics",     # hardware ds_add_f32/f64 and global_atomic_add for the gradient scatter
    "-fno-slp-vectorize",      # the SLP pass pairs independent fp32 ops of neighbouring slice steps into v_pk_* at two v_mov each (more
                               # instructions AND more registers: the benchmark's kernel 128 -> 119 VGPRs without it)
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libdualip_hip.so")


HASH_PATH = LIB_PATH + ".srchash"
# the developer build (tools/ only): the same sources with -DDL_DEVTOOLS -- ablation switches that skip work inside the kernels and the
# tuning constants behind profiles/ become readable from the environment.  Never loaded unless DUALIP_DEV_LIBRARY=1 asks for it (_hip.py).
DEV_LIB_PATH = os.path.join(LIB_DIR, "libdualip_hip_dev.so")
DEV_FLAG = "-DDL_DEVTOOLS"


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
    if not os.path.exists(LIB_PATH) or not os.path.exists(os.path.join(LIB_DIR, "build_manifest.json")):  # (the record of the screens is part of a build)
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


_META_FIELD = __import__("re").compile(r"^  (- |  )(\.[A-Za-z_]+):\s*(.*)$")  # a kernel-level field of .amdgpu_metadata ("  - .x: v" opens an item)
_SREG = __import__("re").compile(r"^s(\d+)$|^s\[(\d+):(\d+)\]$")
# instructions whose FIRST operand is a scalar destination (everything s_* that writes, plus the lane reads)
_NO_SDST = ("s_waitcnt", "s_nop", "s_branch", "s_cbranch", "s_barrier", "s_endpgm", "s_sleep", "s_setprio", "s_cmp", "s_bitcmp", "s_setreg", "s_sendmsg",
            "s_store", "s_buffer_store", "s_dcache", "s_icache", "s_setpc", "s_trap", "s_sethalt", "s_inst_prefetch", "s_clause", "s_code_end", "s_set_gpr",
            "s_ttrace", "s_waitcnt_depctr", "s_delay_alu", "s_wait_", "s_atc_probe", "s_version")


def _sreg(tok):
    """(lo, hi) of a scalar register operand ``sN`` / ``s[a:b]``, else None."""
    m = _SREG.match(tok.strip().rstrip(","))
    if not m:
        return None
    if m.group(1) is not None:
        return int(m.group(1)), int(m.group(1))
    return int(m.group(2)), int(m.group(3))


def _sgpr_pair_defects(asm_path: str):
    """The SECOND code-generation defect seen here (tools/gdb_fault.sh): a 64-bit value that was loaded as part of a
    multi-dword scalar load (``s_load_dwordx4 s[52:55]``: the row stride in s[54:55]) had ONE HALF overwritten by an unrelated scalar
    load (``s_load_dword s55``: gridDim.x) while the pair was still live; both registers were then spilled to adjacent VGPR lanes and
    later reloaded AS A PAIR and used as a 64-bit operand -- 157 * 2^32 + 320 instead of the stride, a memory fault at best.
    The screen walks every kernel's instructions in program order and reports a reload of two adjacent spill lanes into an adjacent
    scalar pair that is then used as ``s[P:P+1]`` when, at spill time, the two source registers had been defined together by one
    multi-register instruction and exactly one of them had since been redefined ALONE by a scalar memory load.  Text order, no
    control-flow analysis: a heuristic that matches the observed miscompile and stays silent on the committed kernels."""
    out, kernel = [], "?"
    lastdef, lastop, group = {}, {}, {}   # per SGPR: index of the defining instruction, its mnemonic, (def index, lo, hi) of a multi-register definition
    slot = {}                             # (vgpr, lane) -> (sreg, def index, op, group) at spill time
    reloads = []                          # (idx, P, vgpr, lane, origin)
    with open(asm_path, errors="replace") as fh:
        lines = fh.read().split("\n")
    instrs = []
    for ln in lines:
        t = ln.strip()
        m = _KERNEL.match(t)
        if m:
            instrs.append(("kernel", m.group(1)))
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if t:
            instrs.append(("i", t))
    pending = []  # suspicious reloaded pairs awaiting a 64-bit use: (P, description, expires at, the STALE register of the pair: P or P + 1)
    for idx, (kind, t) in enumerate(instrs):
        if kind == "kernel":
            kernel = t
            lastdef, lastop, group, slot, pending = {}, {}, {}, {}, []
            continue
        parts = t.replace(",", " ").split()
        op, ops = parts[0], parts[1:]
        # a pending pair used as a 64-bit operand?
        if pending:
            keep = []
            # (a scalar instruction's FIRST operand is its destination: `s_lshl_b64 s[6:7], s[0:1], 2` overwrites s[6:7], it does not read it)
            read_part = " ".join(ops[1:]) if (op.startswith("s_") and not op.startswith(_NO_SDST)) else t
            for P, desc, until, stale in pending:
                if f"s[{P}:{P + 1}]" in read_part:
                    out.append(f"{kernel} {desc}; the pair is then used as s[{P}:{P + 1}] by `{t}`")
                    continue
                if idx < until:
                    keep.append((P, desc, until, stale))
            pending = keep
        if op == "v_writelane_b32" and len(ops) >= 3:
            sr, vreg = _sreg(ops[1]), ops[0]
            if sr and ops[2].isdigit():
                r = sr[0]
                slot[(vreg, int(ops[2]))] = (r, lastdef.get(r), lastop.get(r), group.get(r))
            continue
        if op == "v_readlane_b32" and len(ops) >= 3:
            sr = _sreg(ops[0])
            if sr and ops[2].isdigit():
                P, vreg, lane = sr[0], ops[1], int(ops[2])
                org = slot.get((vreg, lane))
                lo_org = slot.get((vreg, lane - 1))
                # this reload completes an adjacent pair (P-1 <- lane-1 just before, P <- lane)?
                prev = reloads[-1] if reloads else None
                if org and lo_org and prev and prev[1] == P - 1 and prev[2] == vreg and prev[3] == lane - 1 and idx - prev[0] <= 4 and lo_org[0] + 1 == org[0]:
                    (rl, dl, ol, gl), (rh, dh, oh, gh) = lo_org, org
                    for whole, (rx, dx, ox) in ((gl, (rh, dh, oh)), (gh, (rl, dl, ol))):
                        # `whole` = the multi-register definition one half still carries; the OTHER half (rx) was redefined alone, by a scalar load
                        if whole and whole[1] <= rl and rh <= whole[2] and dx is not None and dx != whole[0] and dx > whole[0] and str(ox).startswith(("s_load_dword ", "s_load_dword", "s_buffer_load_dword")) \
                                and not str(ox).startswith(("s_load_dwordx", "s_buffer_load_dwordx")):
                            pending.append((P - 1, f"reloads s[{P - 1}:{P}] from lanes {lane - 1}/{lane} of {vreg}: s{rl}/s{rh} were defined together (registers s[{whole[1]}:{whole[2]}]) "
                                                   f"but s{rx} had been overwritten alone by `{ox}` before the spill", idx + 400, (P - 1) if rx == rl else P))
                reloads.append((idx, P, vreg, lane, org))
                lastdef[P], lastop[P] = idx, "reload"
                group.pop(P, None)
            continue
        if op.startswith("s_") and not op.startswith(_NO_SDST) and ops:
            sr = _sreg(ops[0])
            if sr and pending:
                # A pending pair is dropped when what reaches the 64-bit operand can no longer be "one value's half next to another value's half":
                #  * the SUSPECT half -- the register reloaded from the lane of the half that had been overwritten alone -- is redefined (by anything), or
                #  * the OTHER half is redefined by an EXTENSION of the suspect one: `s_mov_b32 sH, 0` (zero-extension: the lone scalar load WAS the
                #    wanted 32-bit value, e.g. gridDim.x ahead of `s_lshl_b64 s[6:7], s[6:7], 3`) or `s_ashr_i32 sH, sL, 31` (sign extension).
                # Any other redefinition of the other half clears nothing: the suspect half would still ride into the operand (round-5 review:
                # the screen used to drop the report as soon as EITHER half was redefined).
                def _heals(P, stale):
                    other = P + 1 if stale == P else P
                    if sr[0] <= stale <= sr[1]:
                        return True
                    if sr[0] == sr[1] == other:
                        if op == "s_mov_b32" and len(ops) == 2 and ops[1] in ("0", "0x0"):
                            return True
                        if op == "s_ashr_i32" and len(ops) == 3 and ops[1] == f"s{stale}" and ops[2] == "31":
                            return True
                    return False

                pending = [(P, desc, until, stale) for P, desc, until, stale in pending if not _heals(P, stale)]
            if sr:
                for r in range(sr[0], sr[1] + 1):
                    lastdef[r], lastop[r] = idx, op
                    if sr[1] > sr[0]:
                        group[r] = (idx, sr[0], sr[1])
                    else:
                        group.pop(r, None)
        elif op.startswith("v_readfirstlane") and ops:
            sr = _sreg(ops[0])
            if sr:
                lastdef[sr[0]], lastop[sr[0]] = idx, op
                group.pop(sr[0], None)
        elif op.startswith("v_") and ops:
            # vector instructions that WRITE scalar registers: compares in their e64 form (first operand: the lane mask) and the ones
            # with a carry / scale output (second operand) -- a 64-bit mask formed there is a pair defined together
            sr = None
            if op.startswith(("v_cmp", "v_cmpx")):
                sr = _sreg(ops[0])
            elif op.startswith(("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32")) and len(ops) > 1:
                sr = _sreg(ops[1])
            if sr:
                for r in range(sr[0], sr[1] + 1):
                    lastdef[r], lastop[r] = idx, op
                    if sr[1] > sr[0]:
                        group[r] = (idx, sr[0], sr[1])
                    else:
                        group.pop(r, None)
    return out


def _kernel_resources(asm_path: str):
    """{mangled kernel name: {vgpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, scratch_bytes, lds_bytes}} from the
    ``.amdgpu_metadata`` block of one device assembly file (what the code object's notes record)."""
    with open(asm_path, errors="replace") as fh:
        text = fh.read()
    a, b = text.find(".amdgpu_metadata"), text.find(".end_amdgpu_metadata")
    if a < 0 or b < 0:
        return {}
    # The few integer fields wanted here are read line by line: the block is YAML, but PyYAML is not a dependency of this package, and
    # the layout -- one "  - " item per kernel under amdhsa.kernels with its scalar fields four columns in and the nested argument
    # lists deeper -- is what the AMDGPU backend prints for every code-object version hipcc emits.
    body = text[text.index("\n", a) + 1:b]
    k0 = body.find("amdhsa.kernels:")
    if k0 < 0:
        return {}
    items, cur = [], None
    for line in body[k0:].split("\n")[1:]:
        if line and not line.startswith(" "):  # the next top-level key (amdhsa.target, amdhsa.version) or the document's end marker
            break
        m = _META_FIELD.match(line)
        if not m:
            continue
        if m.group(1) == "- ":
            cur = {}
            items.append(cur)
        if cur is not None:
            cur[m.group(2)] = m.group(3).strip().strip("'\"")

    def num(k, key, default=None):
        try:
            return int(k[key])
        except (KeyError, ValueError):
            return default

    out = {}
    for k in items:
        out[k.get(".name", "?")] = {"vgpr_count": num(k, ".vgpr_count"), "agpr_count": num(k, ".agpr_count"), "sgpr_count": num(k, ".sgpr_count"),
                                    "vgpr_spill_count": num(k, ".vgpr_spill_count", 0), "sgpr_spill_count": num(k, ".sgpr_spill_count", 0),
                                    "scratch_bytes": num(k, ".private_segment_fixed_size", 0), "lds_bytes": num(k, ".group_segment_fixed_size", 0)}
    return out


# the benchmark's instantiation -- matching_fused_kernel4<float, unsigned short, LAM_LDS, GRAD_LDS, !HOT, !FAIR, !LANES> -- must not touch
# scratch: a VGPR spill in its hot loop costs the headline several per cent, and its registers are what the second defect went through
BENCHMARK_KERNEL = "_ZN2dl22matching_fused_kernel4IftLb1ELb1ELb0ELb0ELb0EEEvNS_9FusedArgsIT_EE"
MANIFEST_PATH = os.path.join(LIB_DIR, "build_manifest.json")


def _compile_objects(verbose: bool, dev: bool = False):
    """One object per source, compiled in parallel and kept by content hash: editing one file rebuilds one object."""
    from concurrent.futures import ThreadPoolExecutor

    obj_dir = os.path.join(LIB_DIR, "obj_dev" if dev else "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    cflags = [f for f in FLAGS if f != "-shared"] + ([DEV_FLAG] if dev else [])
    jobs, objs = [], []
    for name in SOURCES:
        obj = os.path.join(obj_dir, f"{os.path.splitext(name)[0]}.{_object_hash(name)}.o")
        objs.append(obj)
        if not os.path.exists(obj) or not os.path.exists(obj + ".json"):
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
            found, pairs, resources, scanned = [], [], {}, 0
            for f in sorted(os.listdir(work)):
                if f.endswith(".s") and "amdgcn" in f:
                    scanned += 1
                    found += _spill_defects(os.path.join(work, f))
                    pairs += _sgpr_pair_defects(os.path.join(work, f))
                    resources.update(_kernel_resources(os.path.join(work, f)))
            allow = os.environ.get("DUALIP_BUILD_ALLOW_SPILL_DEFECT", "0") not in ("", "0")
            if not scanned and not allow:  # the screens must not pass because there was nothing to look at
                return name, ("--save-temps=obj left no '*amdgcn*.s' device assembly beside the object (another toolchain's naming?): the code-generation "
                              "screens of dualip_amd/_build.py cannot run; set DUALIP_BUILD_ALLOW_SPILL_DEFECT=1 to build unscreened")
            if found and not allow:
                return name, ("hipcc placed a VGPR spill ahead of the exec restore of a control-flow join (lanes that skipped the branch never store "
                              "their value and reload garbage -- DESIGN.md section 8, tools/spill_exec_check.py):\n  " + "\n  ".join(found) +
                              "\nchange the register pressure of that kernel (or set DUALIP_BUILD_ALLOW_SPILL_DEFECT=1 to build anyway)")
            if pairs and not allow:
                return name, ("hipcc reloads a 64-bit scalar pair from spill lanes after one half was overwritten by an unrelated scalar load (the stride-in-"
                              "s[54:55] miscompile, DESIGN.md section 3.1b):\n  " + "\n  ".join(pairs) +
                              "\nre-read that value from the kernel arguments where it is used (or set DUALIP_BUILD_ALLOW_SPILL_DEFECT=1 to build anyway)")
            bench = resources.get(BENCHMARK_KERNEL)
            if bench and (bench["vgpr_spill_count"] or bench["scratch_bytes"]) and not allow:
                return name, (f"the benchmark instantiation of the fused kernel uses scratch ({bench}): its hot loop must stay in registers "
                              "(DUALIP_BUILD_ALLOW_SPILL_DEFECT=1 builds anyway)")
            with open(tmp + ".json", "w") as fh:
                json.dump({"source": name, "screened_assembly_files": scanned, "spill_before_exec_restore": found, "sgpr_pair_half_redefined": pairs, "kernels": resources}, fh)
            os.replace(tmp + ".json", obj + ".json")
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
        if (f.endswith(".o") and full not in keep) or (f.endswith(".o.json") and full[:-5] not in keep):
            try:
                os.remove(full)
            except OSError:
                pass
    if dev:
        return hipcc, objs
    # what the screens saw, per kernel: registers, spills, scratch (from the code object's metadata) -- checked by tests/test_host_api.py
    manifest = {"flags": FLAGS, "benchmark_kernel": BENCHMARK_KERNEL, "objects": []}
    for obj in objs:
        try:
            with open(obj + ".json") as fh:
                manifest["objects"].append(json.load(fh))
        except OSError:  # an object of an older build of this file (no record beside it): rebuild it next time
            manifest["objects"].append({"source": os.path.basename(obj), "screened_assembly_files": 0, "kernels": {}})
    with open(MANIFEST_PATH + ".tmp", "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    os.replace(MANIFEST_PATH + ".tmp", MANIFEST_PATH)
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
            built_from = source_hash()  # (taken BEFORE compiling: a file edited while hipcc runs must leave the library stale, not fresh)
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
                fh.write(built_from)
            os.replace(HASH_PATH + ".tmp", HASH_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def build_dev(force: bool = False, verbose: bool = False) -> str:
    """libdualip_hip_dev.so: the developer build (DEV_FLAG).  Same screens, its own object cache; rebuilt when the sources changed."""
    os.makedirs(LIB_DIR, exist_ok=True)
    stamp = DEV_LIB_PATH + ".srchash"
    want = source_hash()
    if not force and os.path.exists(DEV_LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return DEV_LIB_PATH
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force:
                shutil.rmtree(os.path.join(LIB_DIR, "obj_dev"), ignore_errors=True)
            hipcc, objs = _compile_objects(verbose, dev=True)
            tmp = f"{DEV_LIB_PATH}.{os.getpid()}.tmp"
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs, "-ldl"], capture_output=True, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc (link, developer build) failed:\n" + r.stdout + r.stderr)
            os.replace(tmp, DEV_LIB_PATH)
            with open(stamp + ".tmp", "w") as fh:
                fh.write(want)
            os.replace(stamp + ".tmp", stamp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return DEV_LIB_PATH


if __name__ == "__main__":
    import sys

    if "--dev" in sys.argv:
        print(build_dev(force="--force" in sys.argv, verbose=True))
    else:
        print(build(force=True, verbose=True))
