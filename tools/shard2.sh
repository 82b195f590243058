#!/bin/bash
# developer aid (round 2): per-rank cost of an 8-GPU run on one GPU -- rank 0's shard of the 100M problem, the exchanged sums
# scaled by 8 in place of the other ranks (bench.py --force-sharded --emulate-world 8), for each exchange back-end and for a
# split shard.  The printed it/s are NOT results.
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 1 --force-sharded --emulate-world 8 --steps 100 --warmup 10 --no-cpu-baseline --no-verify "${@:2}" 2>&1 | tail -1; }
echo "== p2p";          run 29513 --comm p2p
echo "== rccl";         run 29514 --comm rccl
echo "== p2p 2 blocks"; run 29515 --comm p2p --local-blocks 2
echo "== rccl 2 blocks (side-stream overlap)"; run 29516 --comm rccl --local-blocks 2
