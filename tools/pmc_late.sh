#!/bin/bash
# developer aid: instruction / LDS counters of the fused kernel late in a solve (bash tools/pmc_late.sh <proj> <entities> <warmup>)
export TMPDIR=/tmp
P=${1:-simplex}; E=${2:-100000000}; W=${3:-800}
RAW=/tmp/pmc_late; rm -rf $RAW; mkdir -p $RAW
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $RAW/a -o a -- python bench.py --entities $E --proj $P --steps 3 --warmup $W --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("/tmp/pmc_late/a/**/*counter_collection.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "fused" in r["Kernel_Name"]]
    ids=sorted(set(int(r["Dispatch_Id"]) for r in rows))
    first, last = ids[:3], ids[-3:]
    for name, sel in (("first3", first), ("last3", last)):
        agg=collections.defaultdict(list)
        for r in rows:
            if int(r["Dispatch_Id"]) in sel: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(name, {k: round(sum(v)/len(v)/1e6,2) for k,v in agg.items()}, "(millions)")
PY
