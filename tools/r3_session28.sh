#!/bin/bash
# round-3 GPU session 28: K = 32 class (columns of 256 .. 512 non-zeros, two per slice) against HEAD
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/s28
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/s28/pytest.log 2>&1; grep -n "passed\|failed\|^FAILED" gpurun_out/s28/pytest.log
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; a=d['aux']; la=a.get('late') or {}; w=a.get('whole_solve') or {}
        print('$1', 'ms/step %.4f kernel %.4f phys %.3f | late %.4f kernel %.4f | whole it/s %.1f' % (d['ms_per_step'], r['kernel_avg_ms'], r.get('frac',0), la.get('ms_per_step',0), la.get('kernel_avg_ms',0), w.get('iterations_per_s',0)))
"; }
for rep in 1 2 3; do for arm in head tree; do dir=/root/repo; [ $arm = head ] && dir=/root/repo/_ab_head
 ( cd $dir && timeout 600 python bench.py --entities 400000 --sparsity 0.03 --proj simplex --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | line "0.4m_x300_simplex $arm" ) >> gpurun_out/s28/ab.log
 ( cd $dir && timeout 600 python bench.py --entities 250000 --sparsity 0.045 --proj simplex --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | line "0.25m_x450_simplex $arm" ) >> gpurun_out/s28/ab.log
 ( cd $dir && timeout 600 python bench.py --entities 1000000 --sparsity 0.01 --proj simplex --no-cpu-baseline --no-verify --steps 100 --warmup 10 2>/dev/null | line "1m_x100_simplex $arm" ) >> gpurun_out/s28/ab.log
 ( cd $dir && timeout 600 python bench.py --entities 10000000 --no-cpu-baseline --no-verify --steps 200 --warmup 20 2>/dev/null | line "10m_mixed $arm" ) >> gpurun_out/s28/ab.log
done; done; sort gpurun_out/s28/ab.log
cd /tmp
run() { rm -rf /tmp/pm; (cd $1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o t -- python benchmark/movielens_like.py --max-iter 300 > /tmp/pm.log 2>&1); f=$(find /tmp/pm -name "*kernel_stats.csv" | head -1); python3 -c "
import csv,sys
r=[x for x in csv.DictReader(open('$f')) if 'matching_fused' in x['Name']][0]
print('movielens_like $1', '::', 'avg us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))"; }
( for rep in 1 2 3; do run /root/repo/_ab_head; run /root/repo; done ) 2>&1 | grep movielens_like | tee /root/repo/gpurun_out/s28/ml.log
