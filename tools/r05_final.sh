#!/bin/bash
# round 5, last kernel commit: traffic records (keyed on the kernels' source hash), the reference's top sweep sizes under the checker, the
# default line, and the kernel-trace summary of the default command.   bash tools/r05_final.sh   (GPU box; ~12 min)
export TMPDIR=/tmp
i=0
for cfg in "" "--proj simplex" "--entities 1000000 --proj box" "--entities 10000000 --proj simplex" "--entities 10000000"; do
  i=$((i+1))
  timeout 900 python bench.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-late --no-verify --measure-traffic --record-traffic > gpurun_out/r05p_record_$i.json 2> gpurun_out/r05p_record_$i.err
  echo "record $i ($cfg): rc=$?"; python -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/r05p_record_$i.json') if l.startswith('{')][-1]); r=d['roofline']
print(r['traffic'], r['physical_bytes_per_launch'], r['kernel_avg_ms'], round(r['frac'],3), r['traffic_source'][:120])"
done
cp profiles/traffic.json gpurun_out/r05p_traffic.json
for n in ${BIG:-250000000 200000000}; do
  timeout 900 python bench.py --entities $n --steps 10 --warmup 3 --no-cpu-baseline --no-late --no-traffic-fallback > gpurun_out/r05p_${n}_line.json 2> gpurun_out/r05p_${n}.err
  echo "$n: rc=$?"; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r05p_${n}_line.json') if l.startswith('{')][-1]); v=d['aux']['verified']
print(d['ms_per_step'], d['roofline']['kernel_avg_ms'], round(d['roofline']['frac'],3), 'verified', v['ok'], [c['name'][:50] for c in v['checks'] if not c['ok']], 'slab', d['aux']['layout']['slab_bytes'])"
done
( time python bench.py > gpurun_out/r05p_default_line.json 2> gpurun_out/r05p_default.err ) 2>&1 | grep real
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r05p_default_line.json') if l.startswith('{')][-1]); r=d['roofline']
print('default:', d['value'], d['ms_per_step'], r['kernel_avg_ms'], round(r['frac'],4), r['traffic'], r['traffic_source'][:200]); print(r.get('frac_of_read_probe'), r.get('read_probe_beaten_by_kernel'), d['aux']['read_probe_GBps'], (d['aux']['late'] or {}).get('achieved_GBps'), d['aux']['verified']['ok'], d['cpu_baseline'])"
RAW=/tmp/prof_r05p; rm -rf $RAW; mkdir -p $RAW
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o t -- python bench.py --no-cpu-baseline --no-verify --no-traffic-fallback > gpurun_out/r05p_trace_bench.log 2>&1
for f in $(find $RAW -name "*kernel_stats.csv"); do cp $f gpurun_out/r05p_bench_100m_mixed_kernel_stats.csv; head -8 $f | cut -c1-220; done
