#!/bin/bash
# developer aid: FETCH_SIZE of the fused kernel under two values of one environment switch, same box.
# (one counter per pass and a timeout: a FETCH_SIZE + WRITE_SIZE pass hung a box for its whole time limit)
# usage: tools/ab_fetch.sh VAR A B -- bench args...
cd /root/repo; export TMPDIR=/tmp
VAR=$1; A=$2; B=$3; shift 3; [ "$1" == "--" ] && shift
for v in $A $B; do
  rm -rf /tmp/abf; env $VAR=$v timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d /tmp/abf -o p -- python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-late --no-verify > /tmp/abf.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/abf/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "fused" in row.get("Kernel_Name", ""): agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print("$VAR=$v", {k: round(x, 1) for k, x in m.items()}, "bytes(2F+W) = %.4f GB" % ((2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1024 / 1e9), "launches", len(agg.get("FETCH_SIZE", [])))
PY
done
