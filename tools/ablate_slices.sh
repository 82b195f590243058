#!/bin/bash
# developer aid: timing ablations of the ONE-lane slices on the 100M all-simplex map (results wrong on purpose).  The committed kernel honours
# the DUALIP_HIP_ABLATE bits 14 .. 17 only in the K-lane / in-place slices of the second binary; this script expects a scratch tree _ab/abl
# (tools/ab.sh snapshot HEAD abl, then in its csrc/sell.h: `constexpr bool DEVAB = true;`, after `if (... (1 << 16))) act = false;` a line
# `if (ab & (1 << 18)) { act = false; theta = 0; vertex = false; }`, `lam = (ab & (1 << 19)) ? 0.5f + r[t] : lam_of(r[t])` in pass 1, and
# after the two group reductions `if (ab & (1 << 20)) { fx_add(acc, sall, mx, w.scale2); return; }`; rebuild it) -- what profiles/r04j_*, r04l_* ran.
#   bits: 15 no scatter, 16 no Newton passes, 18 no projection, 19 no lambda gather, 20 nothing after pass 1.   ABLATES="0 65536 ..." overrides the list.
export TMPDIR=/tmp
export DUALIP_DEV_LIBRARY=1  # the developer build of the scratch tree (the shipped library never reads DUALIP_HIP_ABLATE)
cd _ab/abl
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']
    print('$1', 'ms/step %.4f kernel %.4f' % (d['ms_per_step'], r['kernel_avg_ms']))
"; }
for rep in 1 2; do
for ab in ${ABLATES:-0 65536 32768 98304 262144 294912}; do
  DUALIP_HIP_ABLATE=$ab timeout 600 python bench.py --proj simplex --steps 30 --warmup 5 --no-verify --no-late --no-cpu-baseline --no-traffic-fallback 2>/dev/null | line "100M simplex ablate=$ab"
done
DUALIP_HIP_ABLATE=0 timeout 600 python bench.py --proj box --steps 30 --warmup 5 --no-verify --no-late --no-cpu-baseline --no-traffic-fallback 2>/dev/null | line "100M box"
done
