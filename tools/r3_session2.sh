#!/bin/bash
# round-3 GPU session 2: causal experiment for the misplaced spill; per-rank cost of the partitions; slice-tail rotation A/B
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/s2
echo "=== bad variant (spill ahead of the exec restore), slices-first in every plan" > gpurun_out/s2/spill.log
( cd _ab_bad && DUALIP_HIP_ABLATE=256 timeout 300 python -m pytest tests/test_gpu_edge_cases.py -q -x -k "65536" --timeout 200 -p no:cacheprovider 2>&1 | grep -v "^  File\|Extension modules" | head -40 ) >> gpurun_out/s2/spill.log 2>&1
echo "=== same source with the wave-uniform table fill (no spill ahead of an exec restore), slices-first in every plan" >> gpurun_out/s2/spill.log
( DUALIP_HIP_ABLATE=256 timeout 300 python -m pytest tests/test_gpu_edge_cases.py -q -k "65536" --timeout 200 2>&1 | tail -5 ) >> gpurun_out/s2/spill.log 2>&1
( DUALIP_HIP_ABLATE=256 timeout 900 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_sell.py tests/test_gpu_fuzz.py -q --timeout 300 2>&1 | tail -5 ) >> gpurun_out/s2/spill.log 2>&1
# per-rank cost, 8-rank partitions of the 100M mixed problem (one GPU holds one rank's shard)
for spec in "contiguous 0" "contiguous 3" "contiguous 4" "contiguous 7" "balanced 0" "reference 0" "reference 7"; do set -- $spec
  timeout 600 python bench.py --force-sharded --emulate-world 8 --partition $1 --emulate-rank $2 --no-cpu-baseline --no-verify --steps 100 --warmup 10 > gpurun_out/s2/emu8_$1_$2.json 2> gpurun_out/s2/emu8_$1_$2.err
done
for t in 0 1; do DUALIP_HIP_SELL_TAIL=$t timeout 600 python bench.py --force-sharded --emulate-world 8 --partition contiguous --emulate-rank 4 --no-cpu-baseline --no-verify --steps 100 --warmup 10 > gpurun_out/s2/tail${t}_c4.json 2>/dev/null
  DUALIP_HIP_SELL_TAIL=$t timeout 600 python bench.py --force-sharded --emulate-world 8 --partition balanced --no-cpu-baseline --no-verify --steps 100 --warmup 10 > gpurun_out/s2/tail${t}_b0.json 2>/dev/null
  DUALIP_HIP_SELL_TAIL=$t timeout 600 python bench.py --entities 10000000 --proj simplex --no-cpu-baseline --no-verify --steps 100 --warmup 10 > gpurun_out/s2/tail${t}_10m_simplex.json 2>/dev/null
done
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/s2/bench_100m.json 2> gpurun_out/s2/bench_100m.err
cat gpurun_out/s2/spill.log | tail -30
