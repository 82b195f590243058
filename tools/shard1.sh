#!/bin/bash
# developer aid: the N>1 code path (distributed objective + RCCL all-reduce) with one rank at a per-rank size of an 8-GPU run,
# plain and under rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
E=${1:-12500000}
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-sharded --entities $E --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700
python bench.py --entities $E --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
RAW=/tmp/prof_shard; rm -rf $RAW; mkdir -p $RAW gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o t -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --force-sharded --entities $E --steps 50 --warmup 10 --no-cpu-baseline > /tmp/ps.log 2>&1
tail -1 /tmp/ps.log | cut -c1-300
for f in $(find $RAW -name "*kernel_stats.csv"); do echo "== $f"; head -14 $f | cut -c1-200; cp $f gpurun_out/shard1_kernel_stats.csv; done
