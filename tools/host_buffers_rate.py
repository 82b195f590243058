"""What a caller who keeps the problem in HOST memory pays ("PCIe-inclusive rate").

The C ABI takes device pointers; the Python objectives also accept CPU tensors (the reference's drivers default to
``host_device="cpu"``, examples/movielens_matching/movies_lens_matching.py:227) and stage them to the GPU once, at construction.  This
script times the same synthetic problem (benchmark/synthetic.py, mixed box / simplex map, the benchmark's solver parameters) twice:
tensors already in HBM, and tensors handed over on the CPU in the reference's own format (int64 CSC indices, pageable memory) with the
result returned on the CPU.  bench.py's ``value`` is the first; the second is reported here only.

    python tools/host_buffers_rate.py [entities] [iterations] > gpurun_out/host_buffers.json
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from benchmark.synthetic import generate_matching_problem
    from dualip_amd.objectives.matching import MatchingInputArgs, MatchingSolverDualObjectiveFunction
    from dualip_amd.optimizers.agd import AcceleratedGradientDescent
    from dualip_amd.projections import create_projection_map

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    m, dev, gamma = 10_000, "cuda:0", 1e-3
    prob = generate_matching_problem(n, m, 0.001, seed=42, device=dev, dtype=torch.float32)
    inp = prob["input_args"]
    half = n // 2
    pm = {**create_projection_map("box", {"lower": 0.0, "upper": 1.0}, None, indices=range(0, half)),
          **create_projection_map("simplex", {"z": 1.0}, None, indices=range(half, n))}
    inp.projection_map = pm

    def solve(args, lam0):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f = MatchingSolverDualObjectiveFunction(matching_input_args=args, gamma=gamma)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        solver = AcceleratedGradientDescent(max_iter=iters, gamma=gamma, initial_step_size=1e-3, max_step_size=1e-1, iteration_callback=False)
        res = solver.maximize(f, lam0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, res

    solve(inp, torch.zeros(m, dtype=torch.float32, device=dev))  # (library load, first launches)
    create_dev, solve_dev, res_dev = solve(inp, torch.zeros(m, dtype=torch.float32, device=dev))

    # the reference's own format on the host: int64 CSC indices, two CSC tensors of the same pattern, pageable memory
    colptr, rowidx = inp.A.ccol_indices().cpu().long(), inp.A.row_indices().cpu().long()
    a_cpu, c_cpu, b_cpu = inp.A.values().cpu(), inp.c.values().cpu(), inp.b_vec.cpu()
    A = torch.sparse_csc_tensor(colptr, rowidx, a_cpu, size=inp.A.shape, check_invariants=False)
    C = torch.sparse_csc_tensor(colptr, rowidx, c_cpu, size=inp.A.shape, check_invariants=False)
    host = MatchingInputArgs(A=A, c=C, projection_map=pm, b_vec=b_cpu, equality_mask=None)
    host_bytes = sum(t.numel() * t.element_size() for t in (colptr, rowidx, a_cpu, c_cpu, b_cpu))
    from dualip_amd import _hip

    def host_run(mode):
        """One construction + solve from the host tensors; mode 'native': dl_stage_to_device (pinned, chunked, indices narrowed on the host),
        'torch': one pageable tensor.to(device) per field (round 5).  Also: device memory in use right after the construction."""
        os.environ["DUALIP_HOST_STAGING"] = mode
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        free0, _ = torch.cuda.mem_get_info()
        log0 = len(_hip.STAGING_LOG)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f = MatchingSolverDualObjectiveFunction(matching_input_args=host, gamma=gamma)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        free1, _ = torch.cuda.mem_get_info()
        solver = AcceleratedGradientDescent(max_iter=iters, gamma=gamma, initial_step_size=1e-3, max_step_size=1e-1, iteration_callback=False)
        res = solver.maximize(f, torch.zeros(m, dtype=torch.float32))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        log = _hip.STAGING_LOG[log0:]
        del f
        return t1 - t0, t2 - t1, res, free0 - free1, log

    host_run("native")  # (pinned pool, first use)
    create_cpu, solve_cpu, res_cpu, resident_native, log_native = host_run("native")
    create_t, solve_t, res_t, resident_torch, _ = host_run("torch")
    os.environ.pop("DUALIP_HOST_STAGING", None)
    link_bytes = sum(r["bytes_link"] for r in log_native) + b_cpu.numel() * b_cpu.element_size()
    stage_call_s = sum(r["seconds"] for r in log_native)

    same = bool(torch.equal(res_cpu.dual_val, res_dev.dual_val.cpu())) and res_cpu.dual_val.device.type == "cpu"
    print(json.dumps({
        "workload": f"synthetic matching, {n} entities x {m} destinations, {prob['nnz']} non-zeros, mixed box/simplex map, fp32, {iters} iterations",
        "in_hbm": {"create_s": round(create_dev, 4), "solve_s": round(solve_dev, 4), "iterations_per_s": round(iters / solve_dev, 1)},
        "host_buffers": {"bytes_handed_over": host_bytes, "create_s": round(create_cpu, 4), "solve_s": round(solve_cpu, 4),
                         "staging_s": round(create_cpu - create_dev, 4), "staging_GBps": round(host_bytes / max(create_cpu - create_dev, 1e-9) / 1e9, 2),
                         "iterations_per_s_solve_only": round(iters / solve_cpu, 1),
                         "iterations_per_s_with_staging": round(iters / (solve_cpu + create_cpu - create_dev), 1),
                         "bytes_over_the_link": link_bytes, "link_GBps_inside_the_staging_calls": round(link_bytes / max(stage_call_s, 1e-9) / 1e9, 2),
                         "staging_calls": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in log_native],
                         "device_memory_after_construction_bytes": int(resident_native),
                         "note": "dl_stage_to_device: int64 row indices narrowed to 16 bits on the host (never in HBM as int64), index arrays shared by A and c cross once"},
        "host_buffers_torch_copy": {"create_s": round(create_t, 4), "staging_s": round(create_t - create_dev, 4), "staging_GBps": round(host_bytes / max(create_t - create_dev, 1e-9) / 1e9, 2),
                                    "iterations_per_s_with_staging": round(iters / (solve_t + create_t - create_dev), 1), "device_memory_after_construction_bytes": int(resident_torch),
                                    "same_dual_bits": bool(torch.equal(res_t.dual_val, res_dev.dual_val.cpu())),
                                    "note": "DUALIP_HOST_STAGING=torch: one pageable tensor.to(device) per field of both CSC tensors (round 5's path)"},
        "same_dual_bits": same,
        "device": torch.cuda.get_device_name(0),
    }))


if __name__ == "__main__":
    main()
