#!/bin/bash
# round-3 GPU session 22: stats kernel with sixteen slab loads in flight + DPP reductions, apply kernel requesting its row ahead of the scalars
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/s22
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/s22/pytest.log 2>&1; grep -n "passed\|failed\|^FAILED" gpurun_out/s22/pytest.log
cd /tmp
run() { rm -rf /tmp/pm; (cd $1 && DUALIP_HIP_FUSE_APPLY=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o t -- python bench.py --entities $3 --proj box --no-cpu-baseline --no-verify --no-late --steps 400 --warmup 40 > /tmp/pm.log 2>&1); f=$(find /tmp/pm -name "*kernel_stats.csv" | head -1); python3 -c "
import csv,sys,json
rows={x['Name'].split('(')[0][-40:]:x for x in csv.DictReader(open('$f')) if 'agd_' in x['Name'] or 'matching_fused' in x['Name']}
d=json.loads([l for l in open('/tmp/pm.log') if l.startswith('{')][-1])
print('$1 fuse=$2 n=$3', 'ms/step %.4f'%d['ms_per_step'], ' | '.join('%s %.2f us'%(k[-28:], float(v['AverageNs'])/1e3) for k,v in rows.items()))"; }
( for rep in 1 2; do for f in 0 1; do run /root/repo/_ab_head $f 1000000; run /root/repo $f 1000000; done; run /root/repo/_ab_head 0 12500000; run /root/repo 0 12500000; done ) 2>&1 | tee /root/repo/gpurun_out/s22/ab.log
