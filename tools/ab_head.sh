#!/bin/bash
# _ab_head: rm -rf _ab_head && mkdir _ab_head && git archive HEAD dualip_amd benchmark bench.py include oracle | tar -x -C _ab_head && (cd _ab_head && python -m dualip_amd._build)   (git-ignored; travels with gpurun)
# developer aid: HEAD (copy under _ab_head/) against the working tree, same box: fused-kernel time on the MovieLens-shaped problem
cd /tmp && export TMPDIR=/tmp
run() { rm -rf /tmp/pm; (cd $1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o t -- python benchmark/movielens_like.py --max-iter 300 > /tmp/pm.log 2>&1); f=$(find /tmp/pm -name "*kernel_stats.csv" | head -1); python3 -c "
import csv,sys
r=[x for x in csv.DictReader(open('$f')) if 'matching_fused' in x['Name']][0]
print('$1', '::', 'avg us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))"; }
for rep in 1 2 3; do run /root/repo/_ab_head; run /root/repo; done
