#!/bin/bash
# round-3 GPU session 21: twelve dual loads in flight in the prologue + step scalars reduced on the DPP unit, against HEAD
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/s21
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/s21/pytest.log 2>&1; grep -n "passed\|failed\|^FAILED" gpurun_out/s21/pytest.log
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; a=d['aux']; la=a.get('late') or {}; w=a.get('whole_solve') or {}
        print('$1', 'ms/step %.4f kernel %.4f | late %.4f kernel %.4f | whole it/s %.1f' % (d['ms_per_step'], r['kernel_avg_ms'], la.get('ms_per_step',0), la.get('kernel_avg_ms',0), w.get('iterations_per_s',0)))
"; }
for rep in 1 2 3; do
  for arm in head tree; do
    dir=/root/repo; [ $arm = head ] && dir=/root/repo/_ab_head
    ( cd $dir && timeout 600 python bench.py --entities 1000000 --proj box --no-cpu-baseline --no-verify --steps 400 --warmup 40 2>/dev/null | line "1m_box $arm" ) >> gpurun_out/s21/ab.log
    ( cd $dir && timeout 600 python bench.py --entities 1000000 --no-cpu-baseline --no-verify --steps 400 --warmup 40 2>/dev/null | line "1m_mixed $arm" ) >> gpurun_out/s21/ab.log
    ( cd $dir && timeout 600 python bench.py --entities 10000000 --no-cpu-baseline --no-verify --steps 200 --warmup 20 2>/dev/null | line "10m_mixed $arm" ) >> gpurun_out/s21/ab.log
    ( cd $dir && timeout 600 python bench.py --entities 10000000 --proj box --no-cpu-baseline --no-verify --steps 200 --warmup 20 2>/dev/null | line "10m_box $arm" ) >> gpurun_out/s21/ab.log
    ( cd $dir && timeout 600 python bench.py --entities 12500000 --no-cpu-baseline --no-verify --steps 200 --warmup 20 2>/dev/null | line "12.5m_mixed $arm" ) >> gpurun_out/s21/ab.log
    ( cd $dir && timeout 600 python bench.py --no-cpu-baseline --no-verify --steps 60 --warmup 10 2>/dev/null | line "100m_mixed $arm" ) >> gpurun_out/s21/ab.log
  done
done
sort gpurun_out/s21/ab.log
