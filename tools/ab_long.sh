#!/bin/bash
# developer aid: fused-kernel time on the MovieLens-shaped problem under the single-column-tile knobs (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp
run() { rm -rf /tmp/pm; env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o t -- python /root/repo/benchmark/movielens_like.py --max-iter 300 > /tmp/pm.log 2>&1; f=$(find /tmp/pm -name "*kernel_stats.csv" | head -1); python3 -c "
import csv,sys
r=[x for x in csv.DictReader(open('$f')) if 'matching_fused' in x['Name']][0]
print('$*', '::', 'avg us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))"; }
for rep in 1 2; do
run DUALIP_HIP_XLONG_MIN=100000000 DUALIP_HIP_NO_SNAKE=1 DUALIP_HIP_ABLATE=8
run DUALIP_HIP_XLONG_MIN=100000000 DUALIP_HIP_NO_SNAKE=1
run DUALIP_HIP_XLONG_MIN=100000000
run DUALIP_HIP_XLONG_MIN=1024 DUALIP_HIP_NO_SNAKE=1
run DUALIP_HIP_XLONG_MIN=1024
run DUALIP_HIP_XLONG_MIN=2048
run DUALIP_HIP_XLONG_MIN=4096
done
