#!/bin/bash
# round 6: statistic-major stats partials -- parity of the traces, the timeline, and config 2 / 10M / the 12.5M shard against HEAD (_ab/base3), same box
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -4
DUALIP_HIP_TIMELINE=1 python tools/timeline.py 1000000 box > gpurun_out/r06k_timeline_1m_box.txt 2>&1; tail -14 gpurun_out/r06k_timeline_1m_box.txt
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']
    print('$1', 'ms/step %.4f kernel %.4f frac %.3f' % (d['ms_per_step'], r['kernel_avg_ms'], r['frac']))
"; }
ROOT=$(pwd)
{
for rep in 1 2 3 4; do
for d in _ab/base3 .; do
  (cd $ROOT/$d && python bench.py --entities 1000000 --proj box --steps 400 --warmup 40 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "1M box tree=$d rep$rep")
  (cd $ROOT/$d && python bench.py --entities 10000000 --steps 100 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "10M mixed tree=$d rep$rep")
  (cd $ROOT/$d && python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 60 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback --no-partition-compare 2>/dev/null | line "12.5M rank/8 steps 11-70 tree=$d rep$rep")
done; done
} 2>&1 | tee gpurun_out/r06k_ab_one_wave_step.txt
