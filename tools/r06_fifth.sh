#!/bin/bash
# round 6, fifth GPU call: what moved the 100M headline by +1 % in the A/B of the fourth call?  base tree / new tree / new tree with the old first-update gain
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_lp.py -q -x --timeout 500 2>&1 | tail -3
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']; L = d['aux']['layout']
    print('$1', 'ms/step %.4f kernel %.4f frac %.3f' % (d['ms_per_step'], r['kernel_avg_ms'], r['frac']))
"; }
ROOT=$(pwd)
{
for rep in 1 2 3 4; do
  (cd $ROOT/_ab/base && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "100M mixed steps 6-25 base rep$rep")
  (cd $ROOT && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "100M mixed steps 6-25 new rep$rep")
  (cd $ROOT && DUALIP_DEV_LIBRARY=1 DUALIP_HIP_BALANCE_GAIN0=0.3 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "100M mixed steps 6-25 new(dev lib, gain0=0.3) rep$rep")
  (cd $ROOT && DUALIP_DEV_LIBRARY=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "100M mixed steps 6-25 new(dev lib, gain0=0.6) rep$rep")
  (cd $ROOT/_ab/base && python bench.py --steps 100 --warmup 100 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "100M mixed steps 101-200 base rep$rep")
  (cd $ROOT && python bench.py --steps 100 --warmup 100 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "100M mixed steps 101-200 new rep$rep")
done
} 2>&1 | tee gpurun_out/r06e_ab_100m.txt
