#!/bin/bash
# developer aid: HEAD (a built copy under _ab_head/, git-ignored, travels with gpurun) against the working tree on the same box.
# prepare here:  rm -rf _ab_head && mkdir _ab_head && git archive HEAD dualip_amd benchmark bench.py include oracle profiles/traffic.json | tar -x -C _ab_head && (cd _ab_head && python -c "from dualip_amd import _build; _build.build()")
# usage: tools/ab_head_bench.sh [reps] -- bench args...
export TMPDIR=/tmp
REPS=${1:-2}; shift; [ "$1" == "--" ] && shift
one() { (cd $1 && shift && python bench.py "$@" --no-cpu-baseline --no-verify 2>/dev/null) | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); a = d['aux']
print('   ms/step %.4f kernel %.4f late ms/step %.4f kernel %.4f whole %.4fs' % (d['ms_per_step'], d['roofline']['kernel_avg_ms'], a['late']['ms_per_step'], a['late']['kernel_avg_ms'], a['whole_solve']['seconds']))"; }
for rep in $(seq 1 $REPS); do for d in /root/repo/_ab_head /root/repo; do echo "$d rep=$rep"; one $d "$@"; done; done
