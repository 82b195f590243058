#!/bin/bash
# _ab_head: rm -rf _ab_head && mkdir _ab_head && git archive HEAD dualip_amd benchmark bench.py include oracle | tar -x -C _ab_head && (cd _ab_head && python -m dualip_amd._build)   (git-ignored; travels with gpurun)
# developer aid: HEAD copy (_ab_head/) against the working tree on the same box, bench.py kernel times
# usage: bash tools/ab_bench.sh "<bench args>" [reps]
A="$1"; R=${2:-3}
one() { (cd $1 && timeout 300 python bench.py $A --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'kernel ms', round(d['roofline']['kernel_avg_ms'],4), 'frac', round(d['roofline']['frac'],3))"); }
for i in $(seq $R); do one /root/repo/_ab_head; one /root/repo; done
