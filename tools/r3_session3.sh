#!/bin/bash
# round-3 GPU session 3: suite with the full-size tests; A/B of the fixed-point scalar sums against HEAD; slices-first in the no-LDS plan
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/s3
( time timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 --durations=12 ) > gpurun_out/s3/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s3/pytest.log
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; a=d['aux']; la=a.get('late') or {}; w=a.get('whole_solve') or {}
        print('$1', 'ms/step %.4f kernel %.4f | late %.4f kernel %.4f | whole it/s %.1f' % (d['ms_per_step'], r['kernel_avg_ms'], la.get('ms_per_step',0), la.get('kernel_avg_ms',0), w.get('iterations_per_s',0)))
"; }
for rep in 1 2; do
  for arm in head tree; do
    dir=/root/repo; [ $arm = head ] && dir=/root/repo/_ab_head
    ( cd $dir && DUALIP_HIP_SELL_TAIL=0 timeout 600 python bench.py --no-cpu-baseline --no-verify --steps 60 --warmup 10 2>/dev/null | line "100m_mixed $arm" ) >> gpurun_out/s3/ab.log
    ( cd $dir && DUALIP_HIP_SELL_TAIL=0 timeout 600 python bench.py --entities 10000000 --no-cpu-baseline --no-verify --steps 200 --warmup 20 2>/dev/null | line "10m_mixed $arm" ) >> gpurun_out/s3/ab.log
    ( cd $dir && DUALIP_HIP_SELL_TAIL=0 timeout 600 python bench.py --entities 10000000 --proj box --no-cpu-baseline --no-verify --steps 200 --warmup 20 2>/dev/null | line "10m_box $arm" ) >> gpurun_out/s3/ab.log
    ( cd $dir && DUALIP_HIP_SELL_TAIL=0 timeout 600 python bench.py --entities 10000000 --proj simplex --no-cpu-baseline --no-verify --steps 200 --warmup 20 2>/dev/null | line "10m_simplex $arm" ) >> gpurun_out/s3/ab.log
    ( cd $dir && DUALIP_HIP_SELL_TAIL=0 timeout 600 python bench.py --entities 1000000 --proj box --no-cpu-baseline --no-verify --steps 400 --warmup 40 2>/dev/null | line "1m_box $arm" ) >> gpurun_out/s3/ab.log
  done
done
for ab in 0 256; do DUALIP_HIP_LDS_MODE=none DUALIP_HIP_ABLATE=$ab timeout 600 python bench.py --entities 10000000 --no-cpu-baseline --no-verify --no-late --steps 50 --warmup 10 2>/dev/null | line "10m_mixed_noLDS ablate=$ab" >> gpurun_out/s3/ab.log; done
for ab in 0 256; do DUALIP_HIP_LDS_MODE=grad DUALIP_HIP_ABLATE=$ab timeout 600 python bench.py --entities 10000000 --no-cpu-baseline --no-verify --no-late --steps 50 --warmup 10 2>/dev/null | line "10m_mixed_gradLDS ablate=$ab" >> gpurun_out/s3/ab.log; done
timeout 900 python bench.py > gpurun_out/s3/bench_100m.json 2> gpurun_out/s3/bench_100m.err
cat gpurun_out/s3/ab.log; tail -25 gpurun_out/s3/pytest.log
