#!/bin/bash
# developer aid: A/B of the slice order (longest first / shortest first) and of the continued cyclic deal, same box
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
one() { python bench.py "$@" --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); a = d['aux']
print('   kernel %.4f late %.4f whole %.4fs' % (d['roofline']['kernel_avg_ms'], a['late']['kernel_avg_ms'], a['whole_solve']['seconds']))"; }
for rep in 1 2 3; do
for ord in desc asc; do for abl in 0 16; do
  echo "order=$ord ablate=$abl rep=$rep"
  for cfg in "--steps 30" "--entities 12500000 --steps 100"; do DUALIP_HIP_SELL_ORDER=$ord DUALIP_HIP_ABLATE=$abl one $cfg; done
done; done; done
