#!/bin/bash
# round 6: the fine-grid 32-bit slabs with wide rows -- tests, then config 2 and the 12.5M shard against 64-bit slabs and round 5's grid, same box
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x --timeout 900 > gpurun_out/r06g_tests.txt 2>&1; tail -30 gpurun_out/r06g_tests.txt
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']; L = d['aux']['layout']
    print('$1', 'ms/step %.4f kernel %.4f frac %.3f slab_bytes %s rows_ok %s wide %s overflows %s verified %s' % (d['ms_per_step'], r['kernel_avg_ms'], r['frac'], L.get('slab_bytes'), L.get('slab_rows_ok'), L.get('slab_wide_rows'), L.get('slab_overflows'), (d['aux'].get('verified') or {}).get('ok')))
"; }
{
for rep in 1 2 3; do
for v in default force 0; do
  if [ $v == default ]; then unset DUALIP_HIP_SLAB32; else export DUALIP_HIP_SLAB32=$v; fi
  python bench.py --entities 1000000 --proj box --steps 400 --warmup 40 --no-cpu-baseline --no-late --no-traffic-fallback 2>/dev/null | line "1M box SLAB32=$v rep$rep"
  python bench.py --entities 2000000 --proj mixed --steps 200 --warmup 40 --no-cpu-baseline --no-late --no-traffic-fallback 2>/dev/null | line "2M mixed SLAB32=$v rep$rep"
done; done
unset DUALIP_HIP_SLAB32
} 2>&1 | tee gpurun_out/r06g_slab_wide_rows_same_box.txt
python bench.py --entities 1000000 --proj box --steps 20 --warmup 5 --no-cpu-baseline --measure-traffic 2>/dev/null | tail -1 > gpurun_out/r06g_config2_line.json
python -c "
import json
d=json.load(open('gpurun_out/r06g_config2_line.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['traffic'], r['kernel_avg_ms'], r['frac'], d['aux']['layout']['slab_wide_rows'], (d['aux'].get('whole_solve') or {}).get('iterations_per_s'), d['aux']['verified']['ok'], d['aux']['cache_resident'])"
