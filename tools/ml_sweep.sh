#!/bin/bash
# developer aid: MovieLens-shaped problem (config 1's shape), fused-kernel time per launch under rocprofv3 for a list of settings of ONE
# (developer switches such as DUALIP_HIP_XLONG_COST10 exist only in the developer build: export DUALIP_DEV_LIBRARY=1 for those)
# environment switch:   bash tools/ml_sweep.sh <tree> VAR v1 v2 ...     (e.g. . DUALIP_HIP_XLONG_COST10 160 100 60)
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$1; V=$2; shift 2
for v in "$@"; do
  rm -rf /tmp/pm; (cd $ROOT/$T && env $V=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o t -- python benchmark/movielens_like.py --max-iter 300 --no-verify > /tmp/pm.log 2>&1)
  f=$(find /tmp/pm -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv
for x in csv.DictReader(open('$f')):
    if 'matching_fused' in x['Name']:
        print('$T $V=$v', 'fused avg us', round(float(x['AverageNs'])/1e3,1), 'min', round(float(x['MinNs'])/1e3,1), 'max', round(float(x['MaxNs'])/1e3,1))"
  grep "iterations/s" /tmp/pm.log
done
