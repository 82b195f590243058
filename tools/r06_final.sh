#!/bin/bash
# round 6, the kernels' last commit: traffic records (keyed on the kernels' source hash) for the five BASELINE shapes, the default line, the kernel-trace
# summary of the default command, the emulated ranks, config 1's shape.   bash tools/r06_final.sh   (GPU box; ~15 min)
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
i=0
for cfg in "" "--proj simplex" "--entities 1000000 --proj box" "--entities 10000000 --proj simplex --gamma-decay" "--entities 10000000"; do
  i=$((i+1))
  timeout 900 python bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline --measure-traffic --record-traffic > gpurun_out/r06f_record_$i.json 2> gpurun_out/r06f_record_$i.err
  echo "record $i ($cfg): rc=$?"; python -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/r06f_record_$i.json') if l.startswith('{')][-1]); r=d['roofline']; la=d['aux'].get('late') or {}; ws=d['aux'].get('whole_solve') or {}
print(d['value'], d['ms_per_step'], r['traffic'], r['physical_bytes_per_launch'], r['kernel_avg_ms'], round(r['frac'],3), 'probe frac', r.get('frac_of_read_probe'), 'late frac', la.get('frac'), 'whole it/s', ws.get('iterations_per_s'), 'verified', (d['aux'].get('verified') or {}).get('ok'), 'slab', d['aux']['layout']['slab_bytes'])"
done
cp profiles/traffic.json gpurun_out/r06f_traffic.json
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r06f_default_line.json 2> gpurun_out/r06f_default.err ) 2>&1 | grep real
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06f_default_line.json') if l.startswith('{')][-1]); r=d['roofline']
print('default:', d['value'], d['ms_per_step'], r['kernel_avg_ms'], round(r['frac'],4), r['traffic'], r['traffic_source'][:200]); print(r.get('frac_of_read_probe'), r.get('read_probe_beaten_by_kernel'), d['aux']['read_probe_GBps'], (d['aux']['late'] or {}).get('achieved_GBps'), d['aux']['verified']['ok'], json.dumps(d['cpu_baseline'])[:600])"
RAW=/tmp/prof_r06f; rm -rf $RAW; mkdir -p $RAW
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-traffic-fallback > gpurun_out/r06f_trace_bench.log 2>&1
for f in $(find $RAW -name "*kernel_stats.csv"); do cp $f gpurun_out/r06f_bench_100m_mixed_kernel_stats.csv; grep -i "fused\|agd_\|balance" $f | cut -c1-200; done
bash tools/emulated_ranks.sh 2>&1 | tee gpurun_out/r06f_emulated_ranks_one_gpu.txt
RAW=/tmp/prof_r06f_ml; rm -rf $RAW; mkdir -p $RAW
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o t -- python benchmark/movielens_like.py --max-iter 300 --no-verify > gpurun_out/r06f_movielens_run.txt 2>&1
for f in $(find $RAW -name "*kernel_stats.csv"); do cp $f gpurun_out/r06f_movielens_kernel_stats.csv; grep -i "fused\|agd_" $f | cut -c1-200; done
grep "iterations/s" gpurun_out/r06f_movielens_run.txt
RAW=/tmp/prof_r06f_c2; rm -rf $RAW; mkdir -p $RAW
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o t -- python bench.py --entities 1000000 --proj box --steps 400 --warmup 40 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback > gpurun_out/r06f_trace_c2.log 2>&1
for f in $(find $RAW -name "*kernel_stats.csv"); do cp $f gpurun_out/r06f_config2_1m_box_kernel_stats.csv; grep -i "fused\|agd_" $f | cut -c1-200; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
