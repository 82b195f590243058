#!/bin/bash
# developer aid: several arms on the same box.  usage: tools/ab_arms.sh reps "DIR|ENV=VAL ..." "DIR|..." -- bench args
export TMPDIR=/tmp
REPS=$1; shift; ARMS=(); while [ "$1" != "--" ]; do ARMS+=("$1"); shift; done; shift
one() { (cd $1 && shift && python bench.py "$@" --no-cpu-baseline --no-verify 2>/dev/null) | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); a = d['aux']
print('   ms/step %.4f kernel %.4f late kernel %.4f whole %.4fs' % (d['ms_per_step'], d['roofline']['kernel_avg_ms'], a['late']['kernel_avg_ms'], a['whole_solve']['seconds']))"; }
for rep in $(seq 1 $REPS); do for arm in "${ARMS[@]}"; do dir=${arm%%|*}; envs=${arm#*|}; echo "[$arm] rep=$rep"; env $envs bash -c "$(declare -f one); one $dir $*"; done; done
