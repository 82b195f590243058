#!/bin/bash
# developer tool: run tools/fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE and print reported / actual per access width
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $ROOT/tools/fetch_calib.hip || exit 1
RAW=/tmp/fetch_calib_raw; rm -rf $RAW; mkdir -p $RAW $ROOT/gpurun_out
cd /tmp && rocprofv3 --output-format csv --pmc FETCH_SIZE -d $RAW -o c -- /tmp/fetch_calib > /dev/null 2>&1
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$RAW/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "stream" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in agg.items():
    name = k.split("stream")[1][:40]
    out[name] = {"reported_KiB_mean": sum(v) / len(v), "actual_KiB": 2**20, "reported_over_actual": sum(v) / len(v) / 2**20, "launches": len(v)}
    print(name, out[name])
json.dump(out, open("$ROOT/gpurun_out/r02_fetch_size_calibration.json", "w"), indent=1)
PY
