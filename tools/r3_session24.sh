#!/bin/bash
# round-3 GPU session 24: the emulated 8-rank shard (balanced, rank 0), start-of-session code (_ab3 = c0e23de) against the tree, same box
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/s24
for rep in 1 2 3; do for arm in old tree; do dir=/root/repo; [ $arm = old ] && dir=/root/repo/_ab3
 ( cd $dir && timeout 600 python bench.py --force-sharded --emulate-world 8 --partition balanced --emulate-rank 0 --no-cpu-baseline --no-verify --steps 100 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); a=d['aux']; w=a.get('whole_solve') or {}
        print('emu8_balanced_0 $arm ms/iter %.4f kernel %.4f late %.4f whole it/s %.0f exch us %.1f' % (d['ms_per_step'], d['roofline']['kernel_avg_ms'], (a.get('late') or {}).get('ms_per_step',0), w.get('iterations_per_s',0), (a.get('collective') or {}).get('us_per_exchange',0)))
" ) >> gpurun_out/s24/ab.log
 ( cd $dir && timeout 600 python bench.py --entities 12500000 --no-cpu-baseline --no-verify --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('12.5m_standalone $arm ms/iter %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_avg_ms']))
" ) >> gpurun_out/s24/ab.log
done; done; sort gpurun_out/s24/ab.log
