#!/bin/bash
# developer aid: the kernel-level numbers of one build -- 10M box / simplex / mixed and the 100M headline, early and late windows
export TMPDIR=/tmp
show() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r, a = d['roofline'], d['aux']
late = a.get('late') or {}
ws = a.get('whole_solve') or {}
print('$1', 'ms/step %.4f' % d['ms_per_step'], 'kernel %.4f' % r['kernel_avg_ms'], 'frac %.3f' % r['frac'], '| late kernel %.4f frac %.3f' % (late.get('kernel_avg_ms', 0), late.get('frac', 0)), '| whole %.3fs %.0f it/s' % (ws.get('seconds', 0), ws.get('iterations_per_s', 0)), '| setup %.2fs' % a['setup_s'], 'slices', a['layout'].get('slices'), 'verified', (a.get('verified') or {}).get('ok'))
"; }
for p in box simplex mixed; do python bench.py --entities 10000000 --proj $p --steps 100 --warmup 5 --no-cpu-baseline --no-traffic-fallback ${EXTRA} 2>/dev/null | show "10M $p"; done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-traffic-fallback ${EXTRA} 2>/dev/null | show "100M mixed"
python bench.py --steps 30 --warmup 5 --proj simplex --no-cpu-baseline --no-traffic-fallback ${EXTRA} 2>/dev/null | show "100M simplex"
