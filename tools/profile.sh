#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile.sh <tag> [bench args...]
# kernel-trace stats pass + two PMC passes of the same bench command; only summaries are kept under gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
RAW=/tmp/prof_raw_$TAG
rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
which rocprofv3
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python bench.py "$@" --no-cpu-baseline --no-verify > $OUT/bench_trace.log 2>&1
echo "trace rc=$?"; tail -2 $OUT/bench_trace.log | cut -c1-600
rocprofv3 --output-format csv --pmc FETCH_SIZE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $RAW/pmc1 -o pmc1 -- python bench.py "$@" --steps 3 --warmup ${PMC_WARMUP:-1} --no-cpu-baseline --no-late --no-verify > $OUT/bench_pmc1.log 2>&1
echo "pmc1 rc=$?"
rocprofv3 --output-format csv --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $RAW/pmc2 -o pmc2 -- python bench.py "$@" --steps 3 --warmup ${PMC_WARMUP:-1} --no-cpu-baseline --no-late --no-verify > $OUT/bench_pmc2.log 2>&1
echo "pmc2 rc=$?"
rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM -d $RAW/pmc3 -o pmc3 -- python bench.py "$@" --steps 3 --warmup ${PMC_WARMUP:-1} --no-cpu-baseline --no-late --no-verify > $OUT/bench_pmc3.log 2>&1
echo "pmc3 rc=$?"
find $RAW -type f | head -30; du -sh $RAW
for f in $(find $RAW/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; echo "== $f"; head -12 $f | cut -c1-300; done
python - <<PY
import csv, glob, collections, json
summary = {}
for tag in ("pmc1","pmc2","pmc3"):
    for f in glob.glob("$RAW/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name","")
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, d in agg.items():
            if "fused" in k or "reduce_partials" in k or "agd_step" in k:
                short = k.split("(")[0][-70:]
                summary.setdefault(short, {}).update({c: {"mean": sum(v)/len(v), "n": len(v)} for c, v in d.items()})
for k, v in summary.items():
    print(k)
    for c, s in v.items():
        print("   %-28s %.6g  (n=%d)" % (c, s["mean"], s["n"]))
json.dump(summary, open("$OUT/pmc_summary.json", "w"), indent=1)
PY
rm -rf $RAW
