#!/bin/bash
# round 6, third GPU call: suite; config 2 timeline with the prologue's stamps; A/B of the round's kernel-side changes against HEAD~ (_ab/base);
# the balance's first-updates gain (developer library); host-buffer staging
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 -rs ) > gpurun_out/r06c_suite.txt 2>&1
tail -6 gpurun_out/r06c_suite.txt
DUALIP_HIP_TIMELINE=1 python tools/timeline.py 1000000 box > gpurun_out/r06c_timeline_1m_box.txt 2>&1; tail -16 gpurun_out/r06c_timeline_1m_box.txt
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
  (cd $ROOT/$d && python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 20 --warmup 5 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback --no-partition-compare 2>/dev/null | line "12.5M rank/8 steps 6-25 tree=$d rep$rep")
done; done
} 2>&1 | tee gpurun_out/r06c_ab_tree.txt
{
for rep in 1 2; do
for cfg in "0.3 8" "0.6 8" "0.6 16" "0.8 16" "0.45 24"; do
  set -- $cfg
  DUALIP_DEV_LIBRARY=1 DUALIP_HIP_BALANCE_GAIN0=$1 DUALIP_HIP_BALANCE_LAUNCHES=$2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$rep bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 20 --warmup 5 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback --no-partition-compare 2>/dev/null | line "12.5M rank/8 steps 6-25 gain0=$1 first=$2 rep$rep"
  DUALIP_DEV_LIBRARY=1 DUALIP_HIP_BALANCE_GAIN0=$1 DUALIP_HIP_BALANCE_LAUNCHES=$2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$rep bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 60 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback --no-partition-compare 2>/dev/null | line "12.5M rank/8 steps 11-70 gain0=$1 first=$2 rep$rep"
  DUALIP_DEV_LIBRARY=1 DUALIP_HIP_BALANCE_GAIN0=$1 DUALIP_HIP_BALANCE_LAUNCHES=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "100M mixed steps 6-25 gain0=$1 first=$2 rep$rep"
done; done
} 2>&1 | tee gpurun_out/r06c_balance_gain.txt
python tools/host_buffers_rate.py 10000000 1000 > gpurun_out/r06c_host_buffers_10m.json 2> gpurun_out/r06c_host_buffers_10m.err; tail -c 2500 gpurun_out/r06c_host_buffers_10m.json; tail -3 gpurun_out/r06c_host_buffers_10m.err
python tools/host_buffers_rate.py 100000000 1000 > gpurun_out/r06c_host_buffers_100m.json 2> gpurun_out/r06c_host_buffers_100m.err; tail -c 2500 gpurun_out/r06c_host_buffers_100m.json; tail -3 gpurun_out/r06c_host_buffers_100m.err
