#!/bin/bash
# developer aid: same-box A/B of the 32-bit gradient slabs (plan switch DUALIP_HIP_SLAB32=0 restores 64-bit slabs in the same binary), three
# alternations per shape: BASELINE config 2 (1M entities, box), the per-rank shard of an 8-GPU run (12.5M entities of the mixed map, sharded
# route with the exchange emulated), 10M mixed.   bash tools/slab_ab.sh > gpurun_out/slab_ab.txt
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']
    print('$1', 'ms/step %.4f  fused kernel %.4f ms  slab_bytes %s' % (d['ms_per_step'], r['kernel_avg_ms'], d['aux'].get('layout', {}).get('slab_bytes')))
"; }
for rep in 1 2 3; do
  for v in 1 0; do
    DUALIP_HIP_SLAB32=$v python bench.py --entities 1000000 --proj box --steps 400 --warmup 40 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "1M box        SLAB32=$v rep$rep"
    DUALIP_HIP_SLAB32=$v python bench.py --entities 10000000 --steps 100 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "10M mixed     SLAB32=$v rep$rep"
    DUALIP_HIP_SLAB32=$v python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 100 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "12.5M rank/8  SLAB32=$v rep$rep"
  done
done
