"""The checker ``bench.py`` runs at the benchmark size, outside every timed region (-> ``aux.verified``), and the full-size GPU tests run
on their configurations (tests/test_gpu_fullsize.py).  It compares the HIP path with ``oracle/`` -- the CPU restatement of the
reference, pinned to the reference's goldens -- and with float64 torch recomputations; it is checking infrastructure like the oracle
itself: nothing in ``dualip_amd/`` imports it, and nothing here is timed or shipped."""
import numpy as np


def verify_at_size(dtype_name, gamma, inp, pm_local, f, local, lam, rank=0, world=1, sharded=False, device="cuda:0", comm_backend=None, length_classes=None,
                   skip_route_check=False):
    """Correctness at the benchmark size (bench.py runs it outside every timed region -> aux.verified; tests/test_gpu_fullsize.py
    runs it on the 10M-entity configurations).  Returns {"ok": bool, "checks": [...]}.

    1. the oracle (oracle/, the CPU restatement pinned to the reference's goldens) on slabs of 5000 columns: one inside
       every projection block, one straddling every block boundary, and the last columns of the arrays (largest offsets);
    2. A x, c.x, sum x^2 recomputed from the returned primal with torch ops (float64, chunked);
    3. N = 1: the sharded route (this shard split into two kernel handles + the exchange) against the single objective;
       N > 1: this library's exchange against torch.distributed's all-reduce of the same local sums, and the duals of all
       ranks bit-identical.

    ``length_classes``: [(lo, hi), ...] -- for every class that has a column of lo <= length <= hi, one more oracle slab of 400
    columns around such a column (shapes whose kernel plan depends on the column length: each plan's columns get checked).
    The names of those checks carry ``length class [lo, hi]``."""
    import torch
    import torch.distributed as dist

    import oracle

    out = {"ok": True, "checks": []}
    m, gamma = local.m, float(gamma)
    npdt = np.float32 if dtype_name == "f32" else np.float64
    tol_x = 2e-4 if dtype_name == "f32" else 1e-9

    def note(name, err, tol):
        good = bool(err <= tol)
        out["checks"].append({"name": name, "err": float(err), "tol": tol, "ok": good})
        out["ok"] = out["ok"] and good

    A, C = inp.A, inp.c
    colptr, rows, a_vals, c_vals = A.ccol_indices(), A.row_indices(), A.values(), C.values()
    n_local = A.shape[1]
    packed = local.calculate_packed(lam, gamma, x_out=local._primal_buffer()).clone()
    x = local._primal_buffer()
    lam_h = lam.cpu().numpy()
    entries = list(pm_local.items())
    bounds = []
    for _, e in entries:
        idx = e.indices
        bounds.append((idx.start, idx.stop) if isinstance(idx, range) else (int(min(idx)), int(max(idx)) + 1))
    slabs = []
    W = 5000
    gsl = torch.Generator().manual_seed(7)
    for q, (lo, hi) in enumerate(bounds):
        if hi - lo > W:
            s0 = lo + int(torch.randint(0, hi - lo - W, (1,), generator=gsl))
            slabs.append((f"inside entry {q} ({entries[q][1].proj_type})", s0, s0 + W))
    for q in range(len(bounds) - 1):
        cut = bounds[q][1]
        if cut == bounds[q + 1][0] and cut - W // 2 >= 0 and cut + W // 2 <= n_local:
            slabs.append((f"straddling the cut between entries {q} and {q + 1}", cut - W // 2, cut + W // 2))
    if n_local > W:
        slabs.append(("last columns of the arrays", n_local - W, n_local))
    if length_classes:
        lens_d = colptr[1:] - colptr[:-1]
        for lo_len, hi_len in length_classes:
            cand = torch.nonzero((lens_d >= lo_len) & (lens_d <= hi_len)).flatten()
            if cand.numel() == 0:
                continue
            j = int(cand[int(torch.randint(0, cand.numel(), (1,), generator=gsl))])
            s0 = max(0, min(j - 200, n_local - 400))
            slabs.append((f"length class [{lo_len}, {hi_len}] (column {j}, {int(lens_d[j])} non-zeros)", s0, min(n_local, s0 + 400)))
    for name, lo, hi in slabs:
        cp = colptr[lo : hi + 1].cpu().numpy().astype(np.int64)
        k0, k1 = int(cp[0]), int(cp[-1])
        cproj = np.full(hi - lo, -1, dtype=np.int32)
        for q, (blo, bhi) in enumerate(bounds):
            a0, a1 = max(lo, blo), min(hi, bhi)
            if a1 > a0:
                cproj[a0 - lo : a1 - lo] = q
        projs = [(e.proj_type, e.proj_params) for _, e in entries]
        _, _, _, xo = oracle.matching_calculate(m, hi - lo, cp - k0, rows[k0:k1].cpu().numpy().astype(np.int64), a_vals[k0:k1].cpu().numpy(),
                                                c_vals[k0:k1].cpu().numpy(), lam_h, gamma, projs, col_proj=cproj, dtype=npdt)
        xs = x[k0:k1].cpu().numpy()
        scale = max(float(np.abs(xo).max()), 1e-30)
        note(f"oracle slab [{lo}, {hi}) {name}, non-zeros [{k0}, {k1})", float(np.abs(xs - xo).max()) / scale, tol_x)
    # 2. the sums, recomputed from the primal
    ax = torch.zeros(m, dtype=torch.float64, device=device)
    cx = torch.zeros((), dtype=torch.float64, device=device)
    xx = torch.zeros((), dtype=torch.float64, device=device)
    step = 1 << 26
    for k0 in range(0, x.numel(), step):
        xs = x[k0 : k0 + step].double()
        ax.index_add_(0, rows[k0 : k0 + step].long(), a_vals[k0 : k0 + step].double() * xs)
        cx += (c_vals[k0 : k0 + step].double() * xs).sum()
        xx += (xs * xs).sum()
    tol_s = 1e-5 if dtype_name == "f32" else 1e-11
    note("A x recomputed from the primal (torch, float64)", float((ax - packed[:m]).abs().max() / ax.abs().max().clamp_min(1e-30)), tol_s)
    note("c.x recomputed from the primal", float((cx - packed[m]).abs() / cx.abs().clamp_min(1e-30)), tol_s)
    note("sum x^2 recomputed from the primal", float((xx - packed[m + 1]).abs() / xx.abs().clamp_min(1e-30)), tol_s)
    # 3. sharded against single / this library's exchange against torch.distributed's
    if skip_route_check:
        pass
    elif not sharded:
        from dualip_amd.objectives.matching import MatchingInputArgs, MatchingSolverDualObjectiveFunctionDistributed

        # two blocks, each with its share of EVERY projection entry (the partition bench.py gives the ranks of an N > 1 run)
        blocks = []
        for part in range(2):
            pos, pmb, A_parts = 0, {}, []
            for q, (blo, bhi) in enumerate(bounds):
                mid = blo + (bhi - blo) // 2
                lo, hi = (blo, mid) if part == 0 else (mid, bhi)
                k0, k1 = int(colptr[lo]), int(colptr[hi])
                sub_ptr = (colptr[lo : hi + 1] - k0)
                A_parts.append((sub_ptr, rows[k0:k1], a_vals[k0:k1], c_vals[k0:k1], hi - lo))
                key, e = entries[q]
                pmb[key] = type(e)(proj_type=e.proj_type, proj_params=e.proj_params, indices=range(pos, pos + hi - lo))
                pos += hi - lo
            ptrs, off = [torch.zeros(1, dtype=colptr.dtype, device=device)], 0
            for sp, r_, a_, c_, w_ in A_parts:
                ptrs.append(sp[1:] + off)
                off += int(r_.numel())
            cp_b = torch.cat(ptrs)
            r_b = torch.cat([t[1] for t in A_parts])
            a_b = torch.cat([t[2] for t in A_parts])
            c_b = torch.cat([t[3] for t in A_parts])
            Ab = torch.sparse_csc_tensor(cp_b, r_b, a_b, size=(m, pos), check_invariants=False)
            Cb = torch.sparse_csc_tensor(cp_b, r_b, c_b, size=(m, pos), check_invariants=False)
            blocks.append(MatchingInputArgs(A=Ab, c=Cb, projection_map=pmb, b_vec=None))
        fd = MatchingSolverDualObjectiveFunctionDistributed(blocks, inp.b_vec, gamma, host_device=device, comm_backend=comm_backend)
        r_sh = fd.calculate(lam, gamma=gamma)
        r_1 = f.calculate(lam, gamma=gamma)
        g1 = r_1.dual_gradient.double()
        note("sharded route (two blocks + exchange, world 1) against the single objective: gradient",
             float((r_sh.dual_gradient.double() - g1).abs().max() / g1.abs().max().clamp_min(1e-30)), 1e-6 if dtype_name == "f32" else 1e-12)
        note("... dual objective", abs(float(r_sh.dual_objective) - float(r_1.dual_objective)) / max(abs(float(r_1.dual_objective)), 1e-30), 1e-6 if dtype_name == "f32" else 1e-12)
        out["sharded_backend"] = fd.communicator().backend if fd.communicator() is not None else "torch.distributed"
        del fd, blocks
    else:
        ours = f.calculate_packed(lam, gamma).clone()
        ref = packed.clone()
        for blk in getattr(f, "more_blocks", []):
            ref += blk.calculate_packed(lam, gamma)
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        note("this library's exchange against torch.distributed all_reduce of the same local sums",
             float((ours - ref).abs().max() / ref.abs().max().clamp_min(1e-30)), 1e-12)
        digest = lam.view(torch.int32 if lam.dtype == torch.float32 else torch.int64).to(torch.int64).sum().double()
        lo_, hi_ = digest.clone(), digest.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        note("duals identical on all ranks (byte checksum spread)", float(hi_ - lo_), 0.0)
    torch.cuda.synchronize()
    return out
