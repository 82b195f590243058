#!/bin/bash
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x --timeout 600 -k "hundred_million" ) 2>&1 | tail -6
DUALIP_HIP_TIMELINE=1 python tools/timeline.py 1000000 box > gpurun_out/r06i_timeline_1m_box.txt 2>&1; tail -18 gpurun_out/r06i_timeline_1m_box.txt
DUALIP_HIP_TIMELINE=1 python tools/timeline.py 12500000 mixed > gpurun_out/r06i_timeline_12m5_mixed.txt 2>&1; tail -18 gpurun_out/r06i_timeline_12m5_mixed.txt
DUALIP_HIP_TIMELINE=1 DUALIP_HIP_FUSE_APPLY=0 python tools/timeline.py 1000000 box > gpurun_out/r06i_timeline_1m_box_nofuse.txt 2>&1; tail -18 gpurun_out/r06i_timeline_1m_box_nofuse.txt
