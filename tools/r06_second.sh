#!/bin/bash
# round 6, second GPU call: the suite with the round's new tests, the slab gate on the benchmark's shapes, the default bench line with the whole-problem CPU legs
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 -rs ) > gpurun_out/r06b_suite.txt 2>&1
tail -8 gpurun_out/r06b_suite.txt
line() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']; L = d['aux']['layout']
    print('$1', 'ms/step %.4f kernel %.4f frac %.3f slab_bytes %s rows_ok %s overflows %s' % (d['ms_per_step'], r['kernel_avg_ms'], r['frac'], L.get('slab_bytes'), L.get('slab_rows_ok'), L.get('slab_overflows')))
"; }
for rep in 1 2; do
for v in default force 0; do
  if [ $v == default ]; then unset DUALIP_HIP_SLAB32; else export DUALIP_HIP_SLAB32=$v; fi
  python bench.py --entities 1000000 --proj box --steps 400 --warmup 40 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback 2>/dev/null | line "1M box SLAB32=$v rep$rep"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 100 --warmup 10 --no-cpu-baseline --no-late --no-verify --no-traffic-fallback --no-partition-compare 2>/dev/null | line "12.5M rank/8 SLAB32=$v rep$rep"
done; done 2>&1 | tee gpurun_out/r06b_slab_gate.txt
unset DUALIP_HIP_SLAB32
( time python bench.py ) > gpurun_out/r06b_bench_default.txt 2>&1
tail -1 gpurun_out/r06b_bench_default.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], json.dumps(d['cpu_baseline'], indent=1)[:3000]); print('verified', d['aux']['verified']['ok'])"
grep real gpurun_out/r06b_bench_default.txt
