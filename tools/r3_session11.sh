#!/bin/bash
# round-3 GPU session 11: where do the +4 us per launch come from?  arms: HEAD copy, tree, tree without the lanes loop in the kernel (_ab2; slices of one lane only)
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/s11
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; a=d['aux']; la=a.get('late') or {}; w=a.get('whole_solve') or {}; lay=a['layout']
        print('$1', 'ms/step %.4f kernel %.4f | late %.4f kernel %.4f | whole it/s %.1f' % (d['ms_per_step'], r['kernel_avg_ms'], la.get('ms_per_step',0), la.get('kernel_avg_ms',0), w.get('iterations_per_s',0)))
"; }
for rep in 1 2 3; do
  for arm in head tree noloop; do
    dir=/root/repo; [ $arm = head ] && dir=/root/repo/_ab_head; [ $arm = noloop ] && dir=/root/repo/_ab2
    ( cd $dir && DUALIP_HIP_SELL_LANES=0 timeout 600 python bench.py --entities 10000000 --no-cpu-baseline --no-verify --steps 200 --warmup 20 2>/dev/null | line "10m_mixed $arm" ) >> gpurun_out/s11/ab.log
    ( cd $dir && DUALIP_HIP_SELL_LANES=0 timeout 600 python bench.py --entities 10000000 --proj box --no-cpu-baseline --no-verify --steps 200 --warmup 20 2>/dev/null | line "10m_box $arm" ) >> gpurun_out/s11/ab.log
    ( cd $dir && DUALIP_HIP_SELL_LANES=0 DUALIP_HIP_FUSE_APPLY=0 timeout 600 python bench.py --entities 1000000 --proj box --no-cpu-baseline --no-verify --steps 400 --warmup 40 2>/dev/null | line "1m_box_nofuse $arm" ) >> gpurun_out/s11/ab.log
  done
done
sort gpurun_out/s11/ab.log
