#!/bin/bash
# usage: bash tools/quick.sh [entities] -- parity subset + per-projection bench lines
N=${1:-10000000}
timeout 900 python -m pytest tests/test_gpu_parity.py -q --timeout 600 -x -k "golden or lds_plans or mixed or simplex_eq or movielens or scala" 2>&1 | tail -4
for P in box simplex mixed; do timeout 300 python bench.py --entities $N --steps 20 --proj $P --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$P', 'it/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'kernel ms', round(d['roofline']['kernel_avg_ms'],4), 'frac', round(d['roofline']['frac'],3), 'layout', d['aux']['layout']['layout'])"; done
