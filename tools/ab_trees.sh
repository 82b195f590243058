#!/bin/bash
# developer aid: same-box A/B of SEVERAL trees, alternating, R repetitions, over a list of bench.py argument sets.
#   bash tools/ab_trees.sh R "treeA treeB ..." "args of set 1" "args of set 2" ...
# (trees are directories holding bench.py + a built dualip_amd, e.g. `.` and _ab/<name> made by tools/ab.sh snapshot)
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=$1; TREES=$2; shift 2
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']; a = d.get('aux', {}); la = a.get('late') or {}; w = a.get('whole_solve') or {}
    print('$1', 'ms/step %.4f kernel %.4f frac %.3f | late ms/step %.4f kernel %.4f | whole %.4fs %.1f it/s | read ceiling %s' % (
        d['ms_per_step'], r['kernel_avg_ms'], r.get('frac', 0), la.get('ms_per_step', 0), la.get('kernel_avg_ms', 0), w.get('seconds', 0), w.get('iterations_per_s', 0), a.get('read_probe_GBps')))
"; }
for args in "$@"; do
  echo "== bench.py $args"
  for i in $(seq $R); do for d in $TREES; do (cd $ROOT/$d && timeout 900 python bench.py $args --no-cpu-baseline --no-traffic-fallback 2>/dev/null | line "$d rep$i"); done; done
done
