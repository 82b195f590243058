#!/bin/bash
# round 6: with the cheaper head, does the folded step pay at larger sizes?  DUALIP_HIP_FUSE_APPLY=0/1 in the same binary, alternating
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']; la = d['aux'].get('late') or {}; ws = d['aux'].get('whole_solve') or {}
    print('$1', 'ms/step %.4f kernel %.4f frac %.3f | late ms/step %s | whole it/s %s' % (d['ms_per_step'], r['kernel_avg_ms'], r['frac'], la.get('ms_per_step'), ws.get('iterations_per_s')))
"; }
{
for rep in 1 2 3; do
for v in 0 1; do
  export DUALIP_HIP_FUSE_APPLY=$v
  python bench.py --entities 10000000 --steps 100 --warmup 10 --no-cpu-baseline --no-verify --no-traffic-fallback 2>/dev/null | line "10M mixed FUSE_APPLY=$v rep$rep"
  python bench.py --entities 10000000 --proj simplex --steps 100 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "10M simplex FUSE_APPLY=$v rep$rep"
  python bench.py --entities 3000000 --steps 200 --warmup 20 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "3M mixed FUSE_APPLY=$v rep$rep"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 60 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback --no-partition-compare 2>/dev/null | line "12.5M rank/8 FUSE_APPLY=$v rep$rep"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-traffic-fallback 2>/dev/null | line "100M mixed FUSE_APPLY=$v rep$rep"
done; done
unset DUALIP_HIP_FUSE_APPLY
} 2>&1 | tee gpurun_out/r06l_fuse_apply_sizes.txt
