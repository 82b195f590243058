#!/bin/bash
# round-3 GPU session 20: the round's profile set with the final code -- default bench line, kernel stats + counters of the headline and of
# the K-lane shape, kernel stats of the MovieLens-shaped problem and of config 2
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/s20
( time timeout 900 python bench.py > gpurun_out/s20/bench_100m_line.json 2> gpurun_out/s20/bench_100m.err ) 2>&1 | tail -3
tail -c 600 gpurun_out/s20/bench_100m_line.json
bash tools/profile.sh r03b_100m_mixed > gpurun_out/s20/profile_100m.log 2>&1; tail -5 gpurun_out/s20/profile_100m.log
bash tools/profile.sh r03b_x40_simplex --entities 2500000 --sparsity 0.004 --proj simplex > gpurun_out/s20/profile_x40.log 2>&1; tail -3 gpurun_out/s20/profile_x40.log
timeout 300 python bench.py --entities 1000000 --proj box --no-cpu-baseline --steps 400 --warmup 40 > gpurun_out/s20/config2_1m_box_line.json 2>/dev/null
timeout 300 python bench.py --entities 10000000 --proj simplex --gamma-decay --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/s20/config3_10m_simplex_decay_line.json 2>/dev/null
timeout 300 python bench.py --entities 2500000 --sparsity 0.004 --proj simplex --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/s20/x40_simplex_line.json 2>/dev/null
cd /tmp; rm -rf /tmp/pm; (rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o t -- python benchmark/movielens_like.py --max-iter 1000 > /root/repo/gpurun_out/s20/movielens.log 2>&1); cp $(find /tmp/pm -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/s20/movielens_kernel_stats.csv; head -6 /root/repo/gpurun_out/s20/movielens_kernel_stats.csv | cut -c1-200; tail -2 /root/repo/gpurun_out/s20/movielens.log
