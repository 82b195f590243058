#!/bin/bash
# developer aid: VALU / SALU instructions of the fused kernel under two values of one environment switch (one counter pass each, with a timeout)
# usage: tools/ab_insts.sh VAR A B -- bench args...
cd /root/repo; export TMPDIR=/tmp
VAR=$1; A=$2; B=$3; shift 3; [ "$1" == "--" ] && shift
for v in $A $B; do
  rm -rf /tmp/abi; env $VAR=$v timeout 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU -d /tmp/abi -o p -- python bench.py "$@" --steps 3 --warmup 12 --no-cpu-baseline --no-late --no-verify > /tmp/abi.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/abi/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "fused" in row.get("Kernel_Name", ""): agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("$VAR=$v", {k: "%.4g (last %.4g)" % (sum(x) / len(x), x[-1]) for k, x in agg.items()}, "launches", len(agg.get("SQ_INSTS_VALU", [])))
PY
done
