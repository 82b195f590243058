#!/bin/bash
# round 6, after the last kernel change: the whole suite, then the measurement set of tools/r06_final.sh again (records are keyed on the kernels' source hash)
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -rs ) > gpurun_out/r06h_suite.txt 2>&1
tail -8 gpurun_out/r06h_suite.txt
bash tools/r06_final.sh
