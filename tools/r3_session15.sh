#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/s15
for ab in 0 4096 8192 12288; do echo "== stamp 1 position $ab"; DUALIP_HIP_ABLATE=$ab timeout 300 python tools/timeline_ml.py 60 2>&1 | grep -v Warning | tail -7; done > gpurun_out/s15/timeline.log 2>&1
echo "== no workgroup columns (XLONG_MIN huge)" >> gpurun_out/s15/timeline.log
DUALIP_HIP_XLONG_MIN=100000 timeout 300 python tools/timeline_ml.py 60 2>&1 | tail -6 >> gpurun_out/s15/timeline.log
echo "== in-place one-column slices off (ablate 1024)" >> gpurun_out/s15/timeline.log
DUALIP_HIP_ABLATE=1024 timeout 300 python tools/timeline_ml.py 60 2>&1 | tail -6 >> gpurun_out/s15/timeline.log
cat gpurun_out/s15/timeline.log
