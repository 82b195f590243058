#!/bin/bash
# developer aid: timing ablations of the 64-wide layout (DUALIP_HIP_ABLATE bits: 1 = skip the scatter, 2 = skip the gather,
# 4 = skip the projection; the 256-wide layout carries no ablation code) -- bash tools/ablate.sh
# The shipped library compiles the ablation branches out: this script runs the DEVELOPER build (python -m dualip_amd._build --dev; _hip.py loads it
# under DUALIP_DEV_LIBRARY=1).
export DUALIP_DEV_LIBRARY=1
run() { tag=$1; shift; env DUALIP_HIP_LAYOUT=1 "$@" timeout 300 python bench.py --entities 10000000 --steps 20 --no-cpu-baseline --proj simplex 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['frac'],3))"; }
for A in 0 4 1 2 3; do run "layout 1, simplex, ablate=$A" DUALIP_HIP_ABLATE=$A; done
