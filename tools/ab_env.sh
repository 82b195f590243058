#!/bin/bash
# developer aid: A/B of one environment switch on the same box.  usage: tools/ab_env.sh VAR A B [reps] -- bench args...
export TMPDIR=/tmp
VAR=$1; A=$2; B=$3; REPS=${4:-3}; shift 4; [ "$1" == "--" ] && shift
one() { python bench.py "$@" --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); a = d['aux']
print('   ms/step %.4f kernel %.4f late ms/step %.4f kernel %.4f whole %.4fs' % (d['ms_per_step'], d['roofline']['kernel_avg_ms'], a['late']['ms_per_step'], a['late']['kernel_avg_ms'], a['whole_solve']['seconds']))"; }
for rep in $(seq 1 $REPS); do for v in $A $B; do echo "$VAR=$v rep=$rep"; env $VAR=$v bash -c "$(declare -f one); one $*"; done; done
