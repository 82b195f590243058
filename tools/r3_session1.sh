#!/bin/bash
# round-3 GPU session 1: suite, bug hunt (slices-first in every plan), flush micro-benchmark, bench lines
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/s1
( time timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 ) > gpurun_out/s1/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s1/pytest.log
for rep in 1 2 3; do DUALIP_HIP_ABLATE=256 timeout 300 python -m pytest tests/test_gpu_edge_cases.py -q -k "65536" --timeout 300 2>&1 | tail -15; done > gpurun_out/s1/bug256.log 2>&1
DUALIP_HIP_ABLATE=256 timeout 600 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_sell.py -q --timeout 300 2>&1 | tail -15 > gpurun_out/s1/bug256_wide.log 2>&1
hipcc --offload-arch=gfx950 -O3 tools/flush_bench.hip -o /tmp/flush_bench > /dev/null 2>&1 && (/tmp/flush_bench 10000; /tmp/flush_bench 10000; /tmp/flush_bench 2000) > gpurun_out/s1/flush.log 2>&1
timeout 900 python bench.py > gpurun_out/s1/bench_100m.json 2> gpurun_out/s1/bench_100m.err
timeout 300 python bench.py --entities 1000000 --proj box --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/s1/bench_1m_box.json 2> gpurun_out/s1/bench_1m_box.err
DUALIP_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --entities 20000000 --steps 20 --warmup 5 --no-late > gpurun_out/s1/bench_2rank.json 2> gpurun_out/s1/bench_2rank.err
for part in contiguous balanced; do timeout 600 python bench.py --force-sharded --emulate-world 8 --partition $part --no-cpu-baseline --no-verify --steps 100 --warmup 10 > gpurun_out/s1/emu8_$part.json 2> gpurun_out/s1/emu8_$part.err; done
tail -3 gpurun_out/s1/pytest.log
