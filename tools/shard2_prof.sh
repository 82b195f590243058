#!/bin/bash
# developer aid: rocprofv3 kernel stats of the emulated 8-rank run (tools/shard2.sh), one back-end ($1 = p2p | rccl)
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
B=${1:-p2p}
RAW=/tmp/prof_shard2_$B; rm -rf $RAW; mkdir -p $RAW gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o t -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $GRAFT_REPO_ROOT/bench.py --gpus 1 --force-sharded --emulate-world 8 --comm $B --steps 100 --warmup 10 --no-cpu-baseline --no-verify --no-late > /tmp/ps2.log 2>&1
tail -1 /tmp/ps2.log | cut -c1-200
for f in $(find $RAW -name "*kernel_stats.csv"); do head -12 $f | cut -c1-220; cp $f $GRAFT_REPO_ROOT/gpurun_out/r02_shard_emu8_${B}_kernel_stats.csv; done
