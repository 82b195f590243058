#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools_profile.sh <tag> [bench args...]
# kernel-trace stats pass + two PMC passes of the same bench command; summaries land in gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py "$@" --no-cpu-baseline > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $OUT/pmc1 -o pmc1 -- python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $OUT/pmc2 -o pmc2 -- python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc2.log 2>&1
find $OUT -name "*.csv" | head -20
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
python - <<PY
import csv, glob, collections
for tag in ("pmc1","pmc2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name","")[:60]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, d in agg.items():
            if "fused" in k or "reduce_partials" in k or "agd_step" in k:
                print(tag, k, {c: (sum(v)/len(v), len(v)) for c, v in d.items()})
PY
