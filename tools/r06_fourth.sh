#!/bin/bash
# round 6, fourth GPU call: the whole suite; timelines after the prologue change; A/B against HEAD~ (_ab/base) incl. the headline
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -rs ) > gpurun_out/r06d_suite.txt 2>&1
tail -12 gpurun_out/r06d_suite.txt
DUALIP_HIP_TIMELINE=1 python tools/timeline.py 1000000 box > gpurun_out/r06d_timeline_1m_box.txt 2>&1; tail -14 gpurun_out/r06d_timeline_1m_box.txt
DUALIP_HIP_TIMELINE=1 python tools/timeline.py 12500000 mixed > gpurun_out/r06d_timeline_12m5_mixed.txt 2>&1; tail -14 gpurun_out/r06d_timeline_12m5_mixed.txt
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']; L = d['aux']['layout']
    print('$1', 'ms/step %.4f kernel %.4f frac %.3f slab_bytes %s' % (d['ms_per_step'], r['kernel_avg_ms'], r['frac'], L.get('slab_bytes')))
"; }
ROOT=$(pwd)
{
for rep in 1 2 3; do
for d in _ab/base .; do
  (cd $ROOT/$d && DUALIP_HIP_SLAB32=0 python bench.py --entities 1000000 --proj box --steps 400 --warmup 40 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "1M box (64-bit slabs) tree=$d rep$rep")
  (cd $ROOT/$d && python bench.py --entities 10000000 --steps 100 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "10M mixed tree=$d rep$rep")
  (cd $ROOT/$d && python bench.py --entities 10000000 --proj simplex --steps 100 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "10M simplex tree=$d rep$rep")
  (cd $ROOT/$d && python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 20 --warmup 5 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback --no-partition-compare 2>/dev/null | line "12.5M rank/8 steps 6-25 tree=$d rep$rep")
  (cd $ROOT/$d && python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$rep bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 60 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback --no-partition-compare 2>/dev/null | line "12.5M rank/8 steps 11-70 tree=$d rep$rep")
  (cd $ROOT/$d && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "100M mixed steps 6-25 tree=$d rep$rep")
done; done
} 2>&1 | tee gpurun_out/r06d_ab_tree.txt
