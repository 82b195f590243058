#!/bin/bash
# what the driver runs at round end, on the committed tree: GPU suite, smoke(), the default bench line
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/final
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/final/pytest.log 2>&1; grep -n "passed\|failed\|^FAILED" gpurun_out/final/pytest.log; tail -3 gpurun_out/final/pytest.log | grep real
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | grep -v Warning | tail -4
( time timeout 900 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/final/bench.json') if l.startswith('{')][-1])
r=d['roofline']
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')})
print('roofline', {k:r[k] for k in ('bound','achieved','peak','unit','frac','traffic')}, 'kernel ms', r['kernel_avg_ms'], 'algorithmic_frac', r.get('algorithmic_frac'))
print('cpu_baseline', {k:d['cpu_baseline'][k] for k in ('value','unit','cores','kind')}, 'verified', d['aux']['verified']['ok'])
PY
