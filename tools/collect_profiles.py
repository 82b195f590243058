"""Developer tool: copy the summaries tools/profile.sh left under gpurun_out/prof_<tag>/ into profiles/ (tracked) and rebuild
profiles/traffic.json.   usage: python tools/collect_profiles.py"""
import csv
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = (("r01f_100m_mixed", "r01f_bench_100m_mixed", "mixed_100000000_1"), ("r01f_10m_simplex", "r01f_bench_10m_simplex", "simplex_10000000_1"),
        ("r01f_10m_box", "r01f_bench_10m_box", "box_10000000_1"))
traffic = {}
for tag, name, key in RUNS:
    src = os.path.join(R, "gpurun_out", f"prof_{tag}")
    rows = list(csv.reader(open(os.path.join(src, "kernel_stats.csv"))))
    csv.writer(open(os.path.join(R, "profiles", f"{name}_kernel_stats.csv"), "w")).writerows([rows[0]] + [r for r in rows[1:] if "dl::" in r[0]])
    shutil.copy(os.path.join(src, "pmc_summary.json"), os.path.join(R, "profiles", f"{name}_pmc.json"))
    d = json.load(open(os.path.join(src, "pmc_summary.json")))
    k = [x for x in d if "fused" in x][0]
    traffic[key] = (2 * d[k]["FETCH_SIZE"]["mean"] + d[k]["WRITE_SIZE"]["mean"]) * 1024.0
    tiles = {"mixed_100000000_1": 4.002e6}.get(key, 400181)
    fused = [r for r in rows[1:] if "fused" in r[0]][0]
    print(f"{name}: fused avg {float(fused[3]) / 1e6:.4f} ms min {float(fused[5]) / 1e6:.4f} ms  HBM {traffic[key] / 1e9:.3f} GB  "
          f"VALU/tile {d[k]['SQ_INSTS_VALU']['mean'] / tiles:.0f} SALU/tile {d[k]['SQ_INSTS_SALU']['mean'] / tiles:.0f} LDS/tile {d[k]['SQ_INSTS_LDS']['mean'] / tiles:.1f}  "
          f"wait {d[k]['SQ_WAIT_ANY']['mean'] / d[k]['SQ_WAVE_CYCLES']['mean']:.2f}")
    for r in rows[1:]:
        if "agd_" in r[0] or "reduce_partials" in r[0]:
            print("    ", r[0][:50], r[1], f"{float(r[3]) / 1e3:.1f} us")
traffic["_note"] = ("HBM bytes per fused-kernel launch = (2*FETCH_SIZE + WRITE_SIZE) KiB from separate rocprofv3 --pmc passes of `python bench.py [--entities N --proj P]` "
                    "(profiles/r01f_bench_*_pmc.json); FETCH_SIZE doubled per the gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md, HBM section). "
                    "Key: {proj}_{entities}_{gpus}.")
json.dump(traffic, open(os.path.join(R, "profiles", "traffic.json"), "w"), indent=1)
