#!/bin/bash
# round-3 GPU session 4: the m-sized side of an iteration -- launches against one persistent launch with grid barriers (gridsync_bench);
# config 2 (1M entities, box) under fewer workgroups / the apply folded into the next fused launch
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/s4
( for m in 10000 2000; do timeout 120 tools/gridsync_bench.bin $m 500; done ) > gpurun_out/s4/gridsync.log 2>&1
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; a=d['aux']; la=a.get('late') or {}; w=a.get('whole_solve') or {}
        print('$1', 'ms/step %.4f kernel %.4f | late %.4f kernel %.4f | whole it/s %.1f' % (d['ms_per_step'], r['kernel_avg_ms'], la.get('ms_per_step',0), la.get('kernel_avg_ms',0), w.get('iterations_per_s',0)))
"; }
for rep in 1 2; do
  for wg in 256 192 128 64; do
    for fa in 0 1; do
      DUALIP_HIP_NUM_WG=$wg DUALIP_HIP_FUSE_APPLY=$fa timeout 300 python bench.py --entities 1000000 --proj box --no-cpu-baseline --no-verify --steps 400 --warmup 40 2>/dev/null | line "1m_box wg=$wg fuse_apply=$fa" >> gpurun_out/s4/c2.log
    done
  done
done
for wg in 256 128; do DUALIP_HIP_NUM_WG=$wg timeout 300 python bench.py --entities 12500000 --no-cpu-baseline --no-verify --steps 200 --warmup 20 2>/dev/null | line "12.5m_mixed wg=$wg" >> gpurun_out/s4/c2.log; done
cat gpurun_out/s4/gridsync.log gpurun_out/s4/c2.log
