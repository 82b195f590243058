#!/bin/bash
# developer aid: same-box A/B of two trees (or of one environment switch), alternating, N repetitions.
#   make a tree of another commit:   bash tools/ab.sh snapshot <git-rev> [name]      -> _ab/<name>/ (git-ignored, travels with gpurun, built here)
#   bench.py A/B of two trees:       bash tools/ab.sh bench <dirA> <dirB> <reps> -- <bench.py args>
#   bench.py A/B of one switch:      bash tools/ab.sh env VAR <a> <b> <reps> -- <bench.py args>
#   MovieLens-shaped kernel time:    bash tools/ab.sh movielens <dirA> <dirB> <reps>
# Every line: tag, ms/step, fused-kernel ms, physical frac, late window, whole solve.
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']; a = d.get('aux', {}); la = a.get('late') or {}; w = a.get('whole_solve') or {}
    print('$1', 'ms/step %.4f kernel %.4f frac %.3f | late ms/step %.4f kernel %.4f | whole %.4fs %.1f it/s | read ceiling %s' % (
        d['ms_per_step'], r['kernel_avg_ms'], r.get('frac', 0), la.get('ms_per_step', 0), la.get('kernel_avg_ms', 0), w.get('seconds', 0), w.get('iterations_per_s', 0), a.get('read_probe_GBps')))
"; }
mode=$1; shift
case $mode in
snapshot)
  rev=$1; name=${2:-$1}; rm -rf "$ROOT/_ab/$name"; mkdir -p "$ROOT/_ab/$name"
  (cd "$ROOT" && git archive "$rev" dualip_amd benchmark bench.py include oracle tests/helpers.py tests/__init__.py | tar -x -C "_ab/$name") && (cd "$ROOT/_ab/$name" && python -m dualip_amd._build) ;;
bench)
  A=$1; B=$2; R=$3; shift 3; [ "$1" == "--" ] && shift
  for i in $(seq $R); do for d in $A $B; do (cd $d && timeout 900 python bench.py "$@" --no-cpu-baseline --no-traffic-fallback 2>/dev/null | line "$d rep$i"); done; done ;;
env)
  V=$1; A=$2; B=$3; R=$4; shift 4; [ "$1" == "--" ] && shift
  for i in $(seq $R); do for v in $A $B; do (cd $ROOT && env $V=$v timeout 900 python bench.py "$@" --no-cpu-baseline --no-traffic-fallback 2>/dev/null | line "$V=$v rep$i"); done; done ;;
movielens)
  A=$1; B=$2; R=$3
  run() { rm -rf /tmp/pm; (cd $1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o t -- python benchmark/movielens_like.py --max-iter 300 --no-verify > /tmp/pm.log 2>&1); f=$(find /tmp/pm -name "*kernel_stats.csv" | head -1); python3 -c "
import csv
for x in csv.DictReader(open('$f')):
    if 'matching_fused' in x['Name'] or 'agd_' in x['Name']:
        print('movielens_like $1', x['Name'].split('(')[0][:60], 'avg us', round(float(x['AverageNs'])/1e3,1), 'min', round(float(x['MinNs'])/1e3,1), 'max', round(float(x['MaxNs'])/1e3,1), 'calls', x['Calls'])"; grep "iterations/s" /tmp/pm.log; }
  for i in $(seq $R); do run $A; run $B; done ;;
*) echo "usage: see the header of tools/ab.sh"; exit 2 ;;
esac
