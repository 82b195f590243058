#!/bin/bash
# round-3 GPU session 23: per-rank cost of the 8-rank run with the round's final code (balanced / contiguous / reference partitions), one box
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/s23
timeout 600 python bench.py --no-cpu-baseline --no-verify --steps 100 --warmup 10 > gpurun_out/s23/bench_100m.json 2>/dev/null
for spec in "balanced -1" "contiguous 0" "contiguous 4" "reference 7"; do set -- $spec
  timeout 600 python bench.py --force-sharded --emulate-world 8 --partition $1 $( [ $2 != -1 ] && echo --emulate-rank $2 ) --no-cpu-baseline --no-verify --steps 100 --warmup 10 > gpurun_out/s23/emu8_$1_$2.json 2>/dev/null
done
python - <<'PY'
import json,glob,os
base=None
for f in ['gpurun_out/s23/bench_100m.json']+sorted(glob.glob('gpurun_out/s23/emu8_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); a=d['aux']; w=a.get('whole_solve') or {}
    if base is None: base=(d['ms_per_step'], w.get('iterations_per_s',0))
    print(os.path.basename(f), 'ms/iter %.4f kernel %.4f late %.4f whole it/s %.0f  speed-up %.2f / %.2f  backend %s' % (d['ms_per_step'], d['roofline']['kernel_avg_ms'], (a.get('late') or {}).get('ms_per_step',0), w.get('iterations_per_s',0), base[0]/d['ms_per_step'], (w.get('iterations_per_s',0)/base[1]) if base[1] else 0, (a.get('collective') or {}).get('backend')))
PY
